"""Differentiable fast path for plain `odeint`.

The reference backpropagates *through* the solver by letting autograd record every eager op of
`_runge_kutta_step`, `_interp_fit` / `_interp_evaluate` and the fixed-grid step functions (with
`_UncheckedAssign`, rk_common.py:31-40, to scatter `f` into the stage buffer).  Here the RK arithmetic runs
in HIP kernels that autograd does not see, so each kernel call becomes ONE autograd node with a hand-written
backward.  All elementwise kernels of include/tdeq_hip.h are linear in their state-sized inputs,

    out = sum_m w_m(s) * X_m ,      s = the time-like scalars of the call (dt, x, slope ...),

so one `torch.autograd.Function` serves all of them:

    grad X_m = w_m * g                       `tdeq_scale_many` (g read once; a weight of exactly 1 passes g through)
    grad s   = sum_m dw_m/ds * <g, X_m>      `tdeq_multi_dot`  (only when a time scalar requires grad)

The forward value is produced by the SAME kernel as in no-grad mode (same rounding), the weights w_m are the
T-rounded coefficients that kernel used, and the controller (error norm, accept/reject, next dt) is outside
the graph exactly as in the reference (`_optimal_step_size` is `@torch.no_grad`, misc.py:85).  The one step size
the reference does differentiate — the first, through `_select_initial_step` (misc.py:36-77 is not under no_grad) —
is differentiable here too: `solvers._InitialStepShadow` records the heuristic's formulas with torch ops next to
the kernel-evaluated values, and the first step's combines, stage times and interpolation carry that graph through
their `dt_shadow` / time shadows.

Time gradients (`t.requires_grad`): host doubles drive the kernels; a "shadow" 0-dim tensor with the autograd
graph back to `t` accompanies each time-like scalar that can carry gradient, and `stitch` gives the tensor
handed to `func` the host value and the shadow's gradient (the reference's `_StitchGradient`, misc.py:348-364).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

Scalar = Optional[torch.Tensor]          # shadow of a time-like scalar (0-dim, requires grad) or None


class _Stitch(torch.autograd.Function):
    """forward: `value`; backward: the gradient goes to `shadow` (cast to its dtype)."""

    @staticmethod
    def forward(ctx, shadow, value):
        ctx.shadow_dtype = shadow.dtype
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.shadow_dtype).reshape(()), None


def stitch(value: torch.Tensor, shadow: Scalar) -> torch.Tensor:
    if shadow is None or not (torch.is_grad_enabled() and shadow.requires_grad):
        return value
    return _Stitch.apply(shadow, value)


class _Spec:
    """Everything `_LinearOp` needs about one kernel call."""
    __slots__ = ("kernels", "launch", "w", "dw", "like", "w_fn")

    def __init__(self, kernels, launch, w, dw, like, w_fn=None):
        self.kernels = kernels      # HipKernels
        self.launch = launch        # launch(out) -> None: runs the forward kernel on detached inputs
        self.w = w                  # [M] weights of the state-sized inputs, as rounded by the kernel
        self.dw = dw                # [S][M] derivatives of the weights wrt each time-like scalar
        self.like = like            # tensor giving shape / dtype / device of the output
        self.w_fn = w_fn            # w_fn(scalars) -> [M] weights as differentiable 0-dim tensors (needed only when the
                                    # weights are not linear in the scalars: second-order gradients, see _backward_with_graph)


class _LinearOp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec: _Spec, n_scalars: int, *args):
        xs = args[n_scalars:]
        out = torch.empty_like(spec.like)
        spec.launch(out)
        ctx.spec, ctx.n_scalars = spec, n_scalars
        ctx.scalar_dtypes = [None if s is None else s.dtype for s in args[:n_scalars]]
        ctx.scalars_saved = any(ctx.needs_input_grad[2:2 + n_scalars])
        if ctx.scalars_saved:
            ctx.save_for_backward(*[s if s is not None else xs[0].new_zeros(()) for s in args[:n_scalars]], *xs)
        return out

    @staticmethod
    def backward(ctx, g):
        if torch.is_grad_enabled() and g.requires_grad:
            return _LinearOp._backward_with_graph(ctx, g)
        spec, ns = ctx.spec, ctx.n_scalars
        kern = spec.kernels
        g = g.contiguous()
        need_x = ctx.needs_input_grad[2 + ns:]
        grads_x: List[Optional[torch.Tensor]] = [None] * len(spec.w)
        outs, ws = [], []
        for m, (w, need) in enumerate(zip(spec.w, need_x)):
            if not need or w == 0.0:
                continue
            if w == 1.0:
                grads_x[m] = g
            else:
                buf = torch.empty_like(g)
                grads_x[m] = buf
                outs.append(buf)
                ws.append(w)
        for lo in range(0, len(outs), 14):                       # TDEQ_MAX_TERMS outputs per launch
            kern.scale_many(outs[lo:lo + 14], g, ws[lo:lo + 14])
        grads_s: List[Optional[torch.Tensor]] = [None] * ns
        if any(ctx.needs_input_grad[2:2 + ns]):
            xs = ctx.saved_tensors[ns:]
            dots = torch.cat([kern.multi_dot(g, [x.detach() for x in xs[lo:lo + 14]])
                              for lo in range(0, len(xs), 14)])
            for i in range(ns):
                if ctx.needs_input_grad[2 + i]:
                    dw = torch.tensor(spec.dw[i], dtype=torch.float64, device=g.device)
                    grads_s[i] = (dw * dots).sum().to(ctx.scalar_dtypes[i])
        return (None, None, *grads_s, *grads_x)

    @staticmethod
    def _backward_with_graph(ctx, g):
        """The same vector-Jacobian product written with differentiable torch ops: taken when the backward pass is
        itself being recorded (`create_graph=True` — second-order gradients through the solver, which the reference's
        eager op graph supports, rk_common.py:31-40).  out = sum_m w_m(s) X_m, so grad X_m = w_m g and
        grad s_i = d/ds_i sum_m w_m(s) Re<g, X_m>.  A weight whose scalar carries a graph keeps the kernel's VALUE and
        takes its derivatives from `spec.w_fn` (the weight as a torch expression of the scalars) or, for weights
        linear in their scalar — every stage combine —, from the constant dw/ds."""
        spec, ns = ctx.spec, ctx.n_scalars
        need_x = ctx.needs_input_grad[2 + ns:]
        need_s = [ctx.scalars_saved and ctx.needs_input_grad[2 + i] for i in range(ns)]
        scalars = list(ctx.saved_tensors[:ns]) if ctx.scalars_saved else []
        xs = ctx.saved_tensors[ns:] if ctx.scalars_saved else ()
        wts: List[object] = list(spec.w)
        dwts: List[List[object]] = [list(row) for row in spec.dw]
        if any(need_s):
            if spec.w_fn is not None:
                # value from the kernel's own (T-rounded) weight, derivatives of every order from the torch expression
                live_w, live_dw = spec.w_fn([s if need else None for s, need in zip(scalars, need_s)])
                relink = lambda v, f: v if not isinstance(f, torch.Tensor) else v + (f - f.detach())
                wts = [relink(w, f) for w, f in zip(spec.w, live_w)]
                dwts = [[relink(v, f) for v, f in zip(row, frow)] for row, frow in zip(spec.dw, live_dw)]
            else:
                for m in range(len(wts)):
                    for i in range(ns):
                        if need_s[i] and spec.dw[i][m] != 0.0:
                            wts[m] = wts[m] + spec.dw[i][m] * (scalars[i] - scalars[i].detach())
        grads_x: List[Optional[torch.Tensor]] = [None] * len(spec.w)
        for m, (wt, need) in enumerate(zip(wts, need_x)):
            if not need or (isinstance(wt, float) and wt == 0.0):
                continue
            if isinstance(wt, float):
                grads_x[m] = g if wt == 1.0 else g * wt
            else:
                grads_x[m] = g * wt.to(g.real.dtype if g.is_complex() else g.dtype)
        grads_s: List[Optional[torch.Tensor]] = [None] * ns
        if any(need_s):
            from ._fallback import real_dot
            dots = [real_dot(g, x) for x in xs]
            for i in range(ns):
                if not need_s[i]:
                    continue
                total = None
                for m, d in enumerate(dots):
                    c = dwts[i][m]
                    if isinstance(c, float) and c == 0.0:
                        continue
                    term = d * c
                    total = term if total is None else total + term
                if total is not None:
                    grads_s[i] = total.to(ctx.scalar_dtypes[i])
        return (None, None, *grads_s, *grads_x)


def _needs_graph(xs: Sequence[torch.Tensor], scalars: Sequence[Scalar]) -> bool:
    if not torch.is_grad_enabled():
        return False
    return any(x.requires_grad for x in xs) or any(s is not None and s.requires_grad for s in scalars)


class Ops:
    """Tensor-returning front end of the elementwise kernels used by the solvers.  Each method computes
    `out` with the HIP kernel; when grad mode is on and an input requires grad, the call is recorded as one
    `_LinearOp` node.  `out=` lets a no-grad call write into an existing buffer (e.g. a solution row); a
    recorded call ignores it and returns a fresh tensor, so callers must use the returned tensor."""

    def __init__(self, kernels, np_dtype):
        self.k = kernels
        self.T = np_dtype

    # -- generic --------------------------------------------------------------------------------------
    def _record(self, launch: Callable, xs: List[torch.Tensor], w: List[float],
                scalars: Sequence[Tuple[Scalar, List[float]]], w_fn=None) -> torch.Tensor:
        """Run `launch` as one autograd node (the kernels only read data pointers, so the inputs need no detach)."""
        spec = _Spec(self.k, launch, w, [dw for _, dw in scalars], xs[0], w_fn=w_fn)
        return _LinearOp.apply(spec, len(scalars), *[s for s, _ in scalars], *xs)

    # -- RK stage combines ------------------------------------------------------------------------------
    def combine(self, y0, ks, coefs, dt: float, dt_shadow: Scalar = None, out=None):
        """y0 + sum_j fl_T(coef_j*dt) k_j   (tdeq_stage_combine)."""
        if not _needs_graph((y0, *ks), (dt_shadow,)):
            if out is None:
                out = torch.empty_like(y0)
            self.k.stage_combine(out, y0, ks, coefs, dt)
            return out
        T = self.T
        cT = [float(T(c)) for c in coefs]
        w = [1.0] + [float(T(T(c) * T(dt))) for c in coefs]
        return self._record(lambda o: self.k.stage_combine(o, y0, ks, coefs, dt), [y0, *ks], w,
                            [(dt_shadow, [0.0] + cT)])

    def fixed_stage(self, mode: int, y0, ks, ws, dt: float, dt_shadow: Scalar = None, out=None):
        """mode 0: y0 + dt * sum_j k_j w_j ; mode 1: y0 + (dt k_0) w_0   (tdeq_fixed_stage)."""
        if not _needs_graph((y0, *ks), (dt_shadow,)):
            if out is None:
                out = torch.empty_like(y0)
            self.k.fixed_stage(mode, out, y0, ks, ws, dt)
            return out
        T = self.T
        wT = [float(T(v)) for v in ws]
        w = [1.0] + [float(T(dt)) * v for v in wT]
        return self._record(lambda o: self.k.fixed_stage(mode, o, y0, ks, ws, dt), [y0, *ks], w,
                            [(dt_shadow, [0.0] + wT)])

    def rk4_stage(self, stage: int, y0, k1, k2, k3, k4, dt: float, dt_shadow: Scalar = None, out=None):
        """3/8-rule stages (tdeq_rk4_38_stage)."""
        ks = [k1, k2, k3, k4][:stage]
        if not _needs_graph((y0, *ks), (dt_shadow,)):
            if out is None:
                out = torch.empty_like(y0)
            self.k.rk4_stage(stage, out, y0, k1, k2, k3, k4, dt)
            return out
        dtT, third = float(self.T(dt)), float(self.T(1.0 / 3.0))
        pattern = {1: [third], 2: [-third, 1.0], 3: [1.0, -1.0, 1.0], 4: [0.125, 0.375, 0.375, 0.125]}[stage]
        w = [1.0] + [dtT * p for p in pattern]
        return self._record(lambda o: self.k.rk4_stage(stage, o, y0, k1, k2, k3, k4, dt), [y0] + ks, w,
                            [(dt_shadow, [0.0] + pattern)])

    # -- Adams–Bashforth(–Moulton) --------------------------------------------------------------------
    def _long_sum(self, xs, ws, scalars=()):
        """weighted_sum over more terms than one launch takes: left to right, the running sum carried with weight 1."""
        cap = 8                                                   # TDEQ_MAX_SUM_TERMS
        acc = self.weighted_sum(xs[:cap], ws[:cap], [(sh, dw[:cap]) for sh, dw in scalars])
        lo = cap
        while lo < len(xs):
            hi = lo + cap - 1
            acc = self.weighted_sum([acc, *xs[lo:hi]], [1.0, *ws[lo:hi]],
                                    [(sh, [0.0, *dw[lo:hi]]) for sh, dw in scalars])
            lo = hi
        return acc

    def adams_predict(self, y0, hist, cb, cm, dt: float, dt_shadow: Scalar = None, db=None, out=None):
        """(y0 + dy, dy, delta) with dy = sum_j T(cb_j) f_j and delta = T(dt) * sum_j T(cm_j) f_j (tdeq_adams_predict);
        `cm is None` (explicit method): only y0 + dy.  `db` = d cb / d dt for the time gradient."""
        implicit = cm is not None
        if not _needs_graph((y0, *hist), (dt_shadow,)):
            y = out if out is not None else torch.empty_like(y0)
            if not implicit:
                self.k.adams_predict(y, y0, hist, cb)
                return y, None, None
            dy, delta = torch.empty_like(y0), torch.empty_like(y0)
            self.k.adams_predict(y, y0, hist, cb, cm, dt, dy_out=dy, delta_out=delta)
            return y, dy, delta
        # recorded path: the same sums through the generic linear nodes (same products, same order => same bits)
        sc = [(dt_shadow, list(db))] if dt_shadow is not None else []
        dy = self._long_sum(list(hist), list(cb), sc)
        y = self.weighted_sum([y0, dy], [1.0, 1.0])
        if not implicit:
            return y, None, None
        sm = self._long_sum(list(hist), list(cm))
        delta = self.weighted_sum([sm], [dt], [(dt_shadow, [1.0])] if dt_shadow is not None else [])
        return y, dy, delta

    def adams_correct(self, plan, y0, f, delta, dy_old, c: float, dt_shadow: Scalar = None, m0: float = 0.0):
        """(y0 + dy, dy) with dy = T(c) f + delta; the convergence census of (dy_old, dy) is left in `plan`
        (tdeq_adams_correct)."""
        if not _needs_graph((y0, f, delta, dy_old), (dt_shadow,)):
            y, dy = torch.empty_like(y0), torch.empty_like(y0)
            self.k.adams_correct(plan, dy, dy_old, y_out=y, f=f, delta=delta, y0=y0, c=c)
            return y, dy
        dy = self.weighted_sum([f, delta], [c, 1.0], [(dt_shadow, [m0, 0.0])] if dt_shadow is not None else [])
        y = self.weighted_sum([y0, dy], [1.0, 1.0])
        self.k.adams_correct(plan, dy.detach(), dy_old.detach(), compute=False)
        return y, dy

    # -- interpolation --------------------------------------------------------------------------------
    def lerp(self, y0, y1, slope: float, slope_shadow: Scalar = None, out=None):
        """y0 + slope (y1 - y0)   (tdeq_lerp)."""
        if not _needs_graph((y0, y1), (slope_shadow,)):
            if out is None:
                out = torch.empty_like(y0)
            self.k.lerp(out, y0, y1, slope)
            return out
        s = float(self.T(slope))
        return self._record(lambda o: self.k.lerp(o, y0, y1, slope), [y0, y1], [1.0 - s, s],
                            [(slope_shadow, [-1.0, 1.0])])

    def weighted_sum(self, xs, ws, scalars: Sequence[Tuple[Scalar, List[float]]] = (), out=None, w_fn=None):
        """sum_m ws_m xs_m (tdeq_weighted_sum); `scalars` = [(shadow, d ws / d scalar)]; `w_fn(scalars) -> (weights,
        [d weights / d scalar_i])` as torch expressions when the weights are NOT linear in the scalars (second-order
        gradients need the curvature: the cubic Hermite basis, see _Spec.w_fn)."""
        if not _needs_graph(xs, [s for s, _ in scalars]):
            if out is None:
                out = torch.empty_like(xs[0])
            self.k.weighted_sum(out, xs, ws)
            return out
        w = [float(self.T(v)) for v in ws]
        return self._record(lambda o: self.k.weighted_sum(o, xs, ws), list(xs), w, list(scalars), w_fn=w_fn)

    def dense_eval(self, y0, y1, k: Sequence[torch.Tensor], mid_idx, mid_coef, dt: float, x: float,
                   dt_shadow: Scalar = None, x_shadow: Scalar = None, out=None):
        """Fused `_interp_fit` + `_interp_evaluate` (tdeq_dense_eval).  As a linear map of its inputs
        (interp.py:17-21, 42-47 expanded):
            y0: 1 + 5x^2 - 14x^3 + 8x^4      y1: -5x^2 + 14x^3 - 8x^4
            f0: dt (x - 4x^2 + 5x^3 - 2x^4)  f1: dt (x^2 - 3x^3 + 2x^4)      k_j: mid_j dt (16x^2 - 32x^3 + 16x^4)"""
        f0, f1 = k[0], k[-1]
        ks_mid = [k[j] for j in mid_idx]
        launch = lambda o: self.k.dense_eval(o, y0, y1, f0, f1, ks_mid, mid_coef, dt, x)
        if not _needs_graph((y0, y1, *k), (dt_shadow, x_shadow)):
            if out is None:
                out = torch.empty_like(y0)
            launch(out)
            return out
        T = self.T
        xv, dtv = float(T(x)), float(T(dt))
        x2, x3, x4 = xv * xv, xv ** 3, xv ** 4
        p_f0, dp_f0 = xv - 4 * x2 + 5 * x3 - 2 * x4, 1 - 8 * xv + 15 * x2 - 8 * x3
        p_f1, dp_f1 = x2 - 3 * x3 + 2 * x4, 2 * xv - 9 * x2 + 8 * x3
        p_m, dp_m = 16 * x2 - 32 * x3 + 16 * x4, 32 * xv - 96 * x2 + 64 * x3
        # unique tensors with merged weights (f0 = k_0 and f1 = k_last usually also carry a mid weight)
        xs: List[torch.Tensor] = [y0, y1]
        w = [1 + 5 * x2 - 14 * x3 + 8 * x4, -5 * x2 + 14 * x3 - 8 * x4]
        dwx = [10 * xv - 42 * x2 + 32 * x3, -10 * xv + 42 * x2 - 32 * x3]
        dwdt = [0.0, 0.0]
        slot = {}

        def add(tensor, wv, dx, ddt):
            key = id(tensor)
            if key not in slot:
                slot[key] = len(xs)
                xs.append(tensor)
                w.append(0.0)
                dwx.append(0.0)
                dwdt.append(0.0)
            i = slot[key]
            w[i] += wv
            dwx[i] += dx
            dwdt[i] += ddt

        add(f0, dtv * p_f0, dtv * dp_f0, p_f0)
        add(f1, dtv * p_f1, dtv * dp_f1, p_f1)
        for j, c in zip(mid_idx, mid_coef):
            cT = float(T(c))
            add(k[j], cT * dtv * p_m, cT * dtv * dp_m, cT * p_m)
        order = list(xs)

        def w_fn(scalars):
            """(weights, [d/d dt, d/d x] of the weights) as torch expressions of (dt, x): polynomial in x, bilinear in
            (dt, x) — same formulas as the floats above."""
            dts, xsd = scalars
            # a shadow carries the GRADIENT of its scalar, not its value (e.g. x's shadow is -anchor / width): the value
            # is the host's, the shadow is attached with zero value
            dt_t = torch.full((), dtv, dtype=torch.float64, device=y0.device)
            x_t = torch.full((), xv, dtype=torch.float64, device=y0.device)
            if dts is not None:
                dt_t = dt_t + (dts - dts.detach()).double()
            if xsd is not None:
                x_t = x_t + (xsd - xsd.detach()).double()
            x2_, x3_, x4_ = x_t * x_t, x_t ** 3, x_t ** 4
            pf0, dpf0 = x_t - 4 * x2_ + 5 * x3_ - 2 * x4_, 1 - 8 * x_t + 15 * x2_ - 8 * x3_
            pf1, dpf1 = x2_ - 3 * x3_ + 2 * x4_, 2 * x_t - 9 * x2_ + 8 * x3_
            pm, dpm = 16 * x2_ - 32 * x3_ + 16 * x4_, 32 * x_t - 96 * x2_ + 64 * x3_
            n_x = len(order)
            zero = torch.zeros((), dtype=torch.float64, device=y0.device)
            ww = [1 + 5 * x2_ - 14 * x3_ + 8 * x4_, -5 * x2_ + 14 * x3_ - 8 * x4_] + [zero] * (n_x - 2)
            wx = [10 * x_t - 42 * x2_ + 32 * x3_, -10 * x_t + 42 * x2_ - 32 * x3_] + [zero] * (n_x - 2)
            wd = [zero, zero] + [zero] * (n_x - 2)

            def put(tensor, val, dx, ddt):
                i = slot[id(tensor)]
                ww[i], wx[i], wd[i] = ww[i] + val, wx[i] + dx, wd[i] + ddt
            put(f0, dt_t * pf0, dt_t * dpf0, pf0)
            put(f1, dt_t * pf1, dt_t * dpf1, pf1)
            for j, c in zip(mid_idx, mid_coef):
                cT_ = float(T(c))
                put(k[j], cT_ * dt_t * pm, cT_ * dt_t * dpm, cT_ * pm)
            return ww, [wd, wx]
        spec = _Spec(self.k, launch, w, [dwdt, dwx], y0, w_fn=w_fn)
        return _LinearOp.apply(spec, 2, dt_shadow, x_shadow, *xs)
