"""`odeint` and the `SOLVERS` plugin table — the drop-in surface of torchdiffeq/_impl/odeint.py:19-108
for the explicit-RK hot path (every explicit RK method of the reference)."""
from __future__ import annotations

import torch

from .misc import check_inputs
from .solvers import (RK4, AdaptiveHeunSolver, Bosh3Solver, Dopri5Solver, Dopri8Solver, Euler, Fehlberg2, Heun2,
                      Heun3, Midpoint, Tsit5Solver)

# method name -> solver class.  Same protocol as the reference's table (odeint.py:19-46):
#   SOLVERS[method](func=..., y0=..., rtol=..., atol=..., **options).integrate(t)
# The names below are every explicit Runge–Kutta method of the reference's table, in its order; the
# reference's other methods (implicit RK / Adams / scipy wrapper) are out of scope.
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
}


def odeint(func, y0, t, *, rtol=1e-7, atol=1e-9, method=None, options=None, event_fn=None):
    """Integrate dy/dt = func(t, y), y(t[0]) = y0, returning y at every time in `t`.

    Same signature, defaults, return layout and error behaviour as the reference `odeint`
    (odeint.py:49-108): `y0` is a Tensor or tuple of Tensors of any shape on a ROCm device, `t` a 1-D
    strictly monotone float Tensor; returns a Tensor `[len(t), *y0.shape]` (tuple of such for tuple
    states) in `y0.dtype` with `y[0] == y0`.  Raises ValueError for an unknown `method`.

    The Runge–Kutta arithmetic runs in hand-written HIP kernels (libtdeq_hip.so); it is not recorded
    by autograd — use `odeint_adjoint` for gradients, and call plain `odeint` under `torch.no_grad()`
    when `func` has parameters that require grad (otherwise NotImplementedError is raised).
    """
    if torch.is_grad_enabled():
        y0_list = y0 if isinstance(y0, tuple) else (y0,)
        if any(isinstance(y_, torch.Tensor) and y_.requires_grad for y_ in y0_list) or \
                (isinstance(t, torch.Tensor) and t.requires_grad):
            raise NotImplementedError(
                "torchdiffeq_amd.odeint does not backpropagate through the solver (the RK arithmetic runs in "
                "HIP kernels outside autograd): use odeint_adjoint for gradients wrt y0 / t / parameters.")
    ci = check_inputs(func, y0, t, rtol, atol, method, options, event_fn, SOLVERS)
    # Runs under the caller's grad mode: if `func` produces tensors that require grad, the wrapped func
    # raises (loudly) instead of returning a silently non-differentiable solution.
    solver = SOLVERS[ci.method](func=ci.func, y0=ci.y0_flat, rtol=ci.rtol, atol=ci.atol, **ci.options)
    solution = solver.integrate(ci.t)
    if ci.layout.is_tuple:
        return ci.layout.unpack(solution, (len(ci.t),))
    return solution.view(len(ci.t), *ci.layout.shapes[0])
