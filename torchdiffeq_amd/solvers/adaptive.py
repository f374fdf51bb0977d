"""Adaptive embedded Runge–Kutta pairs driven from the host: the trial-step loop, the device-resident controller with its
look-ahead first stage, captured trial steps (torchdiffeq/_impl/rk_common.py:161-369, misc.py:36-95, interp.py:1-48)."""
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
from .events import AdaptiveEvents


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


class RKAdaptiveStepsizeODESolver(AdaptiveEvents):
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
        if self._sync is not None and hasattr(self.plan, "want_sums"):
            self.plan.want_sums = True      # (torch-op host path: the fp64 sums are formed only for lock-step sharding)
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
        # (per-element tolerances under the built-in norm, `_vec_fused`: tdeq_error_norm_vec_ctrl forms the fp64 ratio and runs
        #  the controller in the same finalize launch — the look-ahead stage follows as for scalar tolerances)
        self._vec_ctrl = self._vec_fused is not None and hasattr(self.kernels, "error_norm_vec_ctrl")
        n_norm_seg = self.layout.n_seg - (self.norm.n_skip_tail if isinstance(self.norm, BuiltinNorm) else
                                          (self._vec_fused[2] if self._vec_ctrl else 0))
        # (lock-step sharding: only when the collective runs on device buffers — backend nccl = RCCL —, where the sums
        # are all-reduced between the norm's finalize and the controller kernel without leaving the GPU)
        sync_dev = self._sync is not None and self._sync.on_device and y0.device.type == "cuda" \
            and hasattr(self.kernels, "step_controller")
        # (reduced-precision states on the HIP kernels — `whole_row_controller`: the controller launch takes the WHOLE error
        #  row instead of continuing a fused partial sum; stage times in the state's type)
        self._whole_row_ctrl = bool(getattr(self.kernels, "whole_row_controller", False)) and self._fuse is None \
            and func.time_dtype == y0.dtype and dist_sync is None
        device_ctrl = (getattr(self.kernels, "device_controller", True)
                       and (self._fuse is not None or self._whole_row_ctrl or self._vec_ctrl)
                       and (isinstance(self.norm, BuiltinNorm) or self._vec_ctrl)
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
        # "auto" = only where it pays (states up to _GRAPH_AUTO_MAX_ELEMENTS) and silently; the built-in default since r06
        # (_graph._DEFAULT_REQUEST).  A captured func runs in Python only while the graph is being built; what makes that
        # safe without the user vouching for it is `auto`'s net: the side-effect fingerprint around the first eager
        # evaluation, the replay-vs-eager probe of the first captured step, the cache key over every tensor storage and
        # plain value func can be seen to hold, and the one-evaluation re-check of a cached graph at the start of every
        # later solve (_GraphStep._recheck)
        wanted, auto = _graph_request(hip_graph)
        self._graph_auto = auto
        self._graph_explicit = _request_is_explicit(hip_graph)    # refusals warn only where captured steps were asked for
        if wanted and _stream_is_capturing():
            wanted = False          # the caller is capturing a graph of its own around this solve: no nested capture
        # "auto": the side-effect test of a not-yet-known func rides on the solve's first evaluation whether or not THIS
        # solver can capture (a method without a fused error row, a large state): the adjoint's backward solve, which may,
        # asks for the verdict before it evaluates func on its own account (_GraphStep.passed_side_effect_test)
        self._graph_watch = wanted and auto and y0.device.type == "cuda"
        self._auto = None           # auto mode: this solve's policy ("now" / "later" / "never", _GraphStep.auto_policy)
        self._auto_steps = 0
        self._hold_pre = False      # auto mode: the eager step before the switch to replays enqueues no look-ahead stage
        self.hip_graph = wanted and device_ctrl and self._sync is None and y0.device.type == "cuda" \
            and (not self._vec_ctrl or getattr(self.kernels, "vec_ctrl_in_graph", False)) \
            and hasattr(self.kernels, "stage_combine_dev") \
            and self.layout.total <= (_GRAPH_AUTO_MAX_ELEMENTS if auto else _GRAPH_MODE_MAX_ELEMENTS) * \
            (2 if (y0.element_size() == 2 and not auto) else 1)      # (16-bit states: the same bytes; at 2^23 elements the
        #                                                               eager loop is still host-bound — six func dispatches)
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
            # the adjoint norms take their one-element time component as |t|, not as an rms (adjoint.py:250, 273): the
            # same number for fp32 / fp64, one rounding apart for 16-bit states (include/tdeq_hip.h `leading_abs`)
            c.leading_abs = 1 if (getattr(self.norm, "leading_scalar", False) and self.plan.numels[0] == 1) else 0
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

    def _before_integrate(self, t_host: List[float]) -> None:
        t0 = t_host[0]
        self._dt_shadow = None
        # "auto" (the default): while nothing is known about func yet, the evaluation every solve starts with doubles as the
        # side-effect test — an evaluation counter, a cache, dropout show up HERE, before any capture is attempted and
        # before the adjoint's proxy check would evaluate func on its own account
        watch = self._graph_watch and not _GraphStep.status_known(self.func.base_func)
        before = _side_effect_fingerprint(self.func.base_func, self.y0.device) if watch else None
        f0 = self.func.eval(t0, self.y0)
        if watch:
            if _side_effect_fingerprint(self.func.base_func, self.y0.device) != before:
                _GraphStep.refuse_func(self, "evaluating it changed its own attributes, buffers or the device's random-"
                                             "number state (an evaluation counter, a cache, dropout ...), which a replay "
                                             "would not repeat")
                self.hip_graph = False
            else:
                _GraphStep.mark_pure(self.func.base_func)
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
        self._graph_key = None
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
        # r06: per-element tolerances under the built-in norm — the heuristic's three norms are tdeq_init_norms_vec launches
        # (the tolerance vectors as two fp64 streams, ATen's promotion per operation), not ~10 fp64 torch ops each
        vec_init = self._vec_fused is not None and hasattr(kern, "init_norms_vec")

        def vec_norm(sums):
            # max over the components of sqrt(mean) in W = fp64 (misc.py:22-33 on the promoted quotients)
            val = 0.0
            for s_, n_ in list(zip(sums, self._numels))[:len(self._numels) - self._vec_fused[2]]:
                if n_:
                    val = _nan_max(val, math.sqrt(s_ / n_))
            return val
        if vec_init:
            kern.init_norms_vec(plan, 0, y0, f0, y0, self._vec_fused[0], self._vec_fused[1])
        else:
            kern.init_norms(plan, 0, y0, f0, y0)
        s0, s1, bad = self._read_norms()
        self._y_nonfinite = any(b != 0 for b in bad)
        if vec_init:
            d0, d1 = S(vec_norm(s0)), S(vec_norm(s1))
        elif user_norm:
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
        if vec_init:
            kern.init_norms_vec(plan, 1, f1, f0, y0, self._vec_fused[0], self._vec_fused[1])
            s2, _, bad = self._read_norms()
            d2_num = S(vec_norm(s2))
        elif user_norm:
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
            self._graph_key = None      # (the key computed at the first step is not trusted ~100 steps later)
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
        # a norm LAUNCH that can continue the partial error row of the step's last combine: the built-in norm, and per-element
        # tolerances on the fused vector-tolerance kernel (tdeq_error_norm_vec's `err_partial`) — same launch sequence, the
        # tolerance vectors are the only extra streams
        vec_fused = self._vec_fused is not None and not builtin_norm and getattr(kern, "vec_partial", False)
        fused_norm = builtin_norm or vec_fused
        err_partial = None
        err_rem = self._fuse[1:] if self._fuse is not None else None     # (stages, weights) left to the norm kernel
        nograd = not torch.is_grad_enabled()      # no-grad solves (the adjoint's two solves, inference): no graph checks
        carry = self._carry if (fused_norm and (nograd or (plain and not k1.requires_grad))) else None
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
            if i == n_rows - 1 and fsal and self._fuse is not None and fused_norm and \
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
        elif self._fuse is not None and fused_norm and \
                not (torch.is_grad_enabled() and (y0.requires_grad or k[-1].requires_grad)):
            sol = self._c_sol
            y1, err_partial = torch.empty_like(y0), torch.empty_like(y0)
            kern.stage_combine_err(y1, err_partial, y0, [k[j] for j in sol.idx], sol.coef, self._fuse[0], dt_signed)
        else:
            y1 = ops.combine(y0, [k[j] for j in self._c_sol.idx], self._c_sol.coef, dt_signed, dsh_signed)
        f1 = k[-1]

        # ---- error ratio (misc.py:80-82) ----
        err = self._c_err
        vec_ctrl = self._vec_ctrl and not builtin_norm and (err_partial is None or vec_fused)
        use_ctrl = lookahead and ((err_partial is not None and (builtin_norm or vec_ctrl))
                                  or (self._whole_row_ctrl and builtin_norm) or vec_ctrl)
        if use_ctrl:
            ctrl = self._ctrl
            ctrl.t0, ctrl.dt = t0, dt
            tnext = torch.empty(ctrl.n_times, dtype=func.time_dtype, device=y0.device)
            if vec_ctrl and err_partial is not None:
                kern.error_norm_vec_ctrl(self.plan, y0, y1, [k[j] for j in err_rem[0]], err_rem[1], dt_signed,
                                         self._vec_fused[0], self._vec_fused[1], ctrl, tnext, partial=err_partial)
            elif vec_ctrl:
                kern.error_norm_vec_ctrl(self.plan, y0, y1, [k[j] for j in err.idx], err.coef, dt_signed,
                                         self._vec_fused[0], self._vec_fused[1], ctrl, tnext)
            elif err_partial is None:
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
        elif err_partial is not None and builtin_norm:
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
            error_ratio, y1_nonfinite = self._user_norm_ratio(y0, y1, k, dt_signed, err_partial, err_rem)
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

    def _user_norm_ratio(self, y0, y1, k, dt_signed, err_partial=None, err_rem=None):
        """User-supplied `norm` callable (misc.py:80-82 with a custom norm): the kernel materialises
        err/tol (padding zero-filled) and the user's own function reduces it."""
        err = self._c_err
        y0, y1 = y0.detach(), y1.detach()
        if self._vec_fused is not None:
            # per-element tolerances under the built-in norm: err / tol and the per-segment sums in one launch, in the
            # reference's promoted precision (fp64); max over the components of sqrt(mean) as misc.py:22-33
            rtol_v, atol_v, n_skip = self._vec_fused
            if err_partial is not None:       # (the step's last combine already summed the row's leading run)
                self.kernels.error_norm_vec(self.plan, y0, y1, [k[j].detach() for j in err_rem[0]], err_rem[1], dt_signed,
                                            rtol_v, atol_v, partial=err_partial)
            else:
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
