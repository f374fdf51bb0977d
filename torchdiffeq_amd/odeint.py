"""`odeint` and the `SOLVERS` plugin table — the drop-in surface of torchdiffeq/_impl/odeint.py:19-108
for the explicit-RK hot path (every explicit RK method of the reference)."""
from __future__ import annotations

import torch

from ._native import device_guard
from .misc import check_inputs, empty_solution, pack_differentiable, plugin_solver_inputs
from .implicit import (SDIRK2, TRBDF2, GaussLegendre4, GaussLegendre6, ImplicitEuler, ImplicitMidpoint, RadauIIA3,
                       RadauIIA5, Trapezoid)
from .scipy_wrapper import ScipyWrapperODESolver
from .solvers import (RK4, AdamsBashforth, AdamsBashforthMoulton, AdaptiveHeunSolver, Bosh3Solver, Dopri5Solver,
                      Dopri8Solver, Euler, Fehlberg2, Heun2, Heun3, Midpoint, Tsit5Solver)

# method name -> solver class.  Same protocol as the reference's table (odeint.py:19-46):
#   SOLVERS[method](func=..., y0=..., rtol=..., atol=..., **options).integrate(t)
# Every name of the reference's table, in its order: the explicit Runge–Kutta methods (the hot path), the Adams
# multistep methods, the implicit RK methods (matrix-free Broyden on the same kernels, implicit.py) and the
# host-side SciPy bridge.
SOLVERS = {
    "dopri8": Dopri8Solver,
    "dopri5": Dopri5Solver,
    "tsit5": Tsit5Solver,
    "bosh3": Bosh3Solver,
    "fehlberg2": Fehlberg2,
    "adaptive_heun": AdaptiveHeunSolver,
    "euler": Euler,
    "midpoint": Midpoint,
    "heun2": Heun2,
    "heun3": Heun3,
    "rk4": RK4,
    "explicit_adams": AdamsBashforth,
    "implicit_adams": AdamsBashforthMoulton,
    "implicit_euler": ImplicitEuler,
    "implicit_midpoint": ImplicitMidpoint,
    "trapezoid": Trapezoid,
    "radauIIA3": RadauIIA3,
    "gl4": GaussLegendre4,
    "radauIIA5": RadauIIA5,
    "gl6": GaussLegendre6,
    "sdirk2": SDIRK2,
    "trbdf2": TRBDF2,
    "fixed_adams": AdamsBashforthMoulton,      # the reference's backward-compatible alias (odeint.py:41-43)
    "scipy_solver": ScipyWrapperODESolver,     # host-side SciPy bridge, not a HIP path (scipy_wrapper.py)
}


_PROMOTING_METHODS = ("euler", "midpoint", "heun2", "heun3", "rk4", "explicit_adams", "implicit_adams", "fixed_adams")


def _zero_dim_promotion(func, y0, t, method, options, event_fn):
    """A type-promotion artefact of the reference, reproduced: a 0-dim fp32 state on an fp64 time grid.  In the
    explicit fixed-grid and Adams steps `dt` (0-dim fp64) multiplies `f` (0-dim fp32) — two 0-dim tensors promote by
    dtype, so `dy`, `y1` and every later stage are fp64: after the very first evaluation (t and y still fp32) the
    whole solve runs in fp64 and only the stored outputs are rounded to fp32 (solvers.py:104-127, rk_common.py:110-157,
    fixed_adams.py:196-223).  Returned: the func to integrate in fp64, or None when the case does not apply (also with
    step callbacks, which would see the promoted state)."""
    if not (event_fn is None and isinstance(y0, torch.Tensor) and y0.dim() == 0
            and y0.dtype in (torch.float32, torch.complex64)             # complex64 promotes to complex128 the same way
            and isinstance(t, torch.Tensor) and t.dtype == torch.float64 and method in _PROMOTING_METHODS):
        return None
    if getattr(func, "callback_step", None) is not None:
        return None
    first = [True]
    low, wide = y0.dtype, _wide_dtype(y0.dtype)
    perturb = bool((options or {}).get("perturb"))
    increasing = bool(len(t) < 2 or t[1] > t[0])

    def promoted(t_, y_):
        if first[0]:
            first[0] = False
            t32 = t_.to(torch.float32)
            if perturb:
                # the reference perturbs the first evaluation time in the STATE's precision, which is still fp32 there
                # (misc.py:185-196: cast, then nextafter towards the interior of the step); t_ arrives perturbed by one
                # fp64 ulp, which the cast to fp32 rounds away again
                toward = float("inf") if increasing else -float("inf")
                t32 = torch.nextafter(t32, torch.full_like(t32, toward))
            return func(t32, y_.to(low)).to(wide)
        return func(t_, y_)
    return promoted


def _wide_dtype(dtype):
    return torch.complex128 if dtype == torch.complex64 else torch.float64


def odeint(func, y0, t, *, rtol=1e-7, atol=1e-9, method=None, options=None, event_fn=None):
    """Integrate dy/dt = func(t, y), y(t[0]) = y0, returning y at every time in `t`.

    Same signature, defaults, return layout and error behaviour as the reference `odeint`
    (odeint.py:49-108): `y0` is a Tensor or tuple of Tensors of any shape on a ROCm device, `t` a 1-D
    strictly monotone float Tensor; returns a Tensor `[len(t), *y0.shape]` (tuple of such for tuple
    states) in `y0.dtype` with `y[0] == y0`.  Raises ValueError for an unknown `method`.
    With `event_fn(t, y) -> Tensor`, `t` must have two entries (start, direction); the solve stops where the
    event function first crosses zero and `(event_t, solution)` is returned, `solution[-1]` = the state there.

    The Runge–Kutta arithmetic runs in hand-written HIP kernels (libtdeq_hip.so).  With grad mode on and
    `y0`, `t` or parameters of `func` requiring grad, every kernel call is recorded as one autograd node with a
    hand-written backward (autodiff.py), so the result can be backpropagated *through* the solver like the
    reference's; `odeint_adjoint` gives the same gradients in O(1) memory.
    """
    promoted = _zero_dim_promotion(func, y0, t, method, options, event_fn)
    if promoted is not None:
        return odeint(promoted, y0.to(_wide_dtype(y0.dtype)), t, rtol=rtol, atol=atol, method=method,
                      options=options).to(y0.dtype)
    ci = check_inputs(func, y0, t, rtol, atol, method, options, event_fn, SOLVERS)
    y0_flat = ci.y0_flat
    if sum(ci.layout.numels) == 0 and ci.event_fn is None:
        return empty_solution(ci, y0_flat)
    if torch.is_grad_enabled():
        y0_list = y0 if ci.layout.is_tuple else (y0,)
        if any(y_.requires_grad for y_ in y0_list):
            y0_flat = pack_differentiable(ci.layout, y0_list)        # backprop through the solver (autodiff.py)
    with device_guard(y0_flat.device):       # kernels go to the state's device, whatever the caller's current one is
        solver_cls = SOLVERS[ci.method]
        # (a class someone registered in SOLVERS — plugin protocol, odeint.py:19-46 — gets inputs it can read)
        options, rtol, atol = plugin_solver_inputs(solver_cls, ci.layout, ci.options, ci.rtol, ci.atol, y0_flat.device)
        solver = solver_cls(func=ci.func, y0=y0_flat, rtol=rtol, atol=atol, **options)
        if ci.event_fn is None:
            solution = solver.integrate(ci.t)
        else:
            event_t, solution = solver.integrate_until_event(ci.t[0], ci.event_fn)
            event_t = event_t.to(ci.t)
            if ci.t_is_reversed:
                event_t = -event_t
    if ci.layout.is_tuple:
        solution = ci.layout.unpack(solution, (len(ci.t),))
    else:
        # rows as the solver returned them (len(t) for every solver of this package; the SciPy wrapper hands back fewer
        # when solve_ivp gives up early — the reference passes that through, odeint.py:98-101)
        solution = solution.view(solution.shape[0], *ci.layout.shapes[0])
    if ci.event_fn is None:
        return solution
    return event_t, solution


def odeint_dense(func, y0, t0, t1, *, rtol=1e-7, atol=1e-9, method=None, options=None):
    """Solve from t0 to t1 and return `dense_output_fn(t_eval) -> y(t_eval)`, the solver's piecewise quartic
    dense output (odeint.py:111-157; dopri5 and tensor states only, like the reference).

    Wire format kept by the returned function: `times[n_steps + 1]` (host) and the per-step interpolation
    coefficients `[n_steps, 5, numel]` = [e, d, c, b, a] written by `tdeq_interp_fit`; an evaluation is one
    `tdeq_weighted_sum` launch over the 5 planes of the step that contains `t_eval`."""
    assert torch.is_tensor(y0)
    t = torch.tensor([t0, t1]).to(t0)
    with torch.no_grad(), device_guard(y0.device):
        ci = check_inputs(func, y0, t, rtol, atol, method, options, None, SOLVERS)
        assert ci.method == "dopri5"
        solver = Dopri5Solver(func=ci.func, y0=ci.y0_flat, rtol=ci.rtol, atol=ci.atol, **ci.options)
        times, coeffs = solver.integrate_dense(ci.t)
    shape, sign, kernels = ci.layout.shapes[0], (-1.0 if ci.t_is_reversed else 1.0), solver.kernels
    np_dtype = solver.np_dtype
    import bisect

    def dense_output_fn(t_eval):
        if isinstance(t_eval, torch.Tensor) and t_eval.numel() != 1:
            # (the reference's closure indexes its coefficient stack with the whole `searchsorted` result: odeint.py:151-156)
            raise IndexError("odeint_dense: the dense output is evaluated at one time per call, got {} times"
                             .format(t_eval.numel()))
        ts = sign * float(t_eval)
        idx = bisect.bisect_right(times, ts)
        if idx >= len(times):
            raise IndexError("index {} is out of bounds for dimension 0 with size {}".format(idx, len(times)))
        ta, tb = times[idx - 1], times[idx]          # idx == 0 wraps to the last entry, as the reference's `times[idx - 1]` does
        assert ta <= ts <= tb, "invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}".format(ta, ts, tb)
        w = solver._w                       # time arithmetic in W = promote_types(options['dtype'], T), then cast to T (interp.py:39-40)
        x = np_dtype(w(w(ts - ta) / w(tb - ta)))
        w, xp = [1.0, float(x)], x
        for _ in range(3):
            xp = np_dtype(xp * x)
            w.append(float(xp))
        planes = coeffs[idx - 1]
        out = torch.empty(planes.shape[1], dtype=planes.dtype, device=planes.device)
        with device_guard(planes.device):
            kernels.weighted_sum(out, list(planes.unbind(0)), w)
        return out.view(shape)

    dense_output_fn.times = times
    dense_output_fn.interp_coeffs = coeffs
    return dense_output_fn


class _NonFiniteAnchor(torch.autograd.Function):
    """A real 0-dim zero that depends on a NON-FINITE event state: what `odeint_event` adds to the event time of a
    trajectory that left the finite range, so that the time stays in the autograd graph; every gradient through it is
    NaN, which is what differentiating through that state gives in the reference (odeint.py:195-231)."""

    @staticmethod
    def forward(ctx, y):
        ctx.like = y
        return torch.zeros((), dtype=y.real.dtype if y.is_complex() else y.dtype, device=y.device)

    @staticmethod
    def backward(ctx, g):
        return torch.full_like(ctx.like, float("nan"))


def odeint_event(func, y0, t0, *, event_fn, reverse_time=False, odeint_interface=odeint, **kwargs):
    """Solve from `t0` until `event_fn(t, y)` crosses zero; returns `(event_t, solution)` whose gradients see the
    event time as a function of the trajectory (same contract as odeint.py:160-231: parameters of the event
    function must be part of the state to receive gradients).

    The solve itself returns a detached event time t* and the state y* = y(t*) attached to the solve's graph.  The
    event is defined by c(t*, y(t*)) = 0, so by the implicit function theorem a perturbation dy of the trajectory at
    t* moves the event time by  dt* = -(dc/dy . dy) / (dc/dt + dc/dy . f)  and the state at the (moved) event by
    dy + f dt*.  Both are attached here as first-order corrections with value ZERO on the flat state,

        t_out = t* - <c_y, y - stop_grad(y)> / (c_t + <c_y, f>)          y_out = y + f (t_out - t*),

    built from ordinary differentiable tensor ops — no dedicated autograd node: autograd derives the backward
    (dL/dy = g_y - c_y (g_t + <g_y, f>) / (c_t + <c_y, f>)) from these two lines."""
    t0_row = t0.reshape(-1)
    t = torch.cat([t0_row, t0_row.detach() + (-1.0 if reverse_time else 1.0)])
    event_t, solution = odeint_interface(func, y0, t, event_fn=event_fn, **kwargs)

    # flat-state views of func and event_fn in (ascending) solver time (dummy tolerances: nothing is solved here)
    ci = check_inputs(func, y0, t, 0.0, 0.0, None, None, event_fn, SOLVERS)
    layout, flat_func, flat_event = ci.layout, ci.func, ci.event_fn
    y_event = pack_differentiable(layout, [s[-1] for s in solution]) if layout.is_tuple else solution[-1].reshape(-1)
    time_sign = -1.0 if reverse_time else 1.0
    ts_value = (event_t.detach() * time_sign) if reverse_time else event_t.detach()      # t* in solver time

    if not (torch.is_grad_enabled() and y_event.requires_grad):
        # nothing to differentiate (forward-only call, no_grad, detached state): the solve's own (event_t, solution) is
        # the answer — no extra evaluation of func, no backward through event_fn (the reference does both only in its
        # autograd Function's backward, odeint.py:195-231)
        return event_t, solution

    # first-order data at the event: f(t*, y*), dc/dt, dc/dy (all constants of the correction below)
    y_const = y_event.detach()
    if not bool(torch.isfinite(y_const).all()):
        # a trajectory that has left the finite range (a diverged training run): the reference still hands back the
        # time its bisection ended at (a NaN sign "differs" from every sign) and the non-finite state; a first-order
        # correction through NaNs would only turn the time into NaN as well.  The added zero keeps event_t in the graph
        # (gradients through it are NaN, as the reference's).
        return event_t + _NonFiniteAnchor.apply(y_event).to(event_t.dtype), solution
    with torch.no_grad(), device_guard(y_const.device):
        nfe_before = flat_func.nfe
        f_event = flat_func(ts_value, y_const)
        flat_func.nfe = nfe_before
    with torch.enable_grad():
        ts_leaf, y_leaf = ts_value.clone().requires_grad_(True), y_const.clone().requires_grad_(True)
        c = flat_event(ts_leaf, y_leaf)
        if c.requires_grad:
            c_t, c_y = torch.autograd.grad(c, (ts_leaf, y_leaf), torch.ones_like(c), allow_unused=True)
        else:               # an event function without a graph (detached, integer / boolean based, a constant)
            c_t = c_y = None
    c_t = torch.zeros_like(ts_value) if c_t is None else c_t
    c_y = torch.zeros_like(y_const) if c_y is None else c_y
    # rate of change of c along the trajectory; the tiny offset keeps a grazing event (rate 0) finite, as the
    # reference's guard does
    rate = c_t + torch.dot(c_y, f_event) + 1e-12

    ts_out = ts_value - torch.dot(c_y, y_event - y_const) / rate
    y_out = y_event + f_event * (ts_out - ts_value)
    event_t = (ts_out * time_sign if reverse_time else ts_out).reshape(event_t.shape).to(event_t.dtype)

    if layout.is_tuple:
        solution = tuple(torch.cat([s[:-1], part[None]], dim=0) for s, part in zip(solution, layout.unpack(y_out)))
    else:
        solution = torch.cat([solution[:-1], y_out.view(layout.shapes[0])[None]], dim=0)
    return event_t, solution
