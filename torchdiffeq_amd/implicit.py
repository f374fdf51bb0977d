"""Fixed-grid implicit Runge–Kutta solvers (`implicit_euler`, `implicit_midpoint`, `trapezoid`, `radauIIA3`, `gl4`,
`radauIIA5`, `gl6`, `sdirk2`, `trbdf2`) — rk_common.py:378-558 + fixed_grid_implicit.py of the reference.

The reference solves the stage equations  K_i = f(t_i, y0 + dt sum_j beta_ij K_j)  with Broyden's ("good") method
starting from J = I and KEEPS THE DENSE MATRIX: `J = torch.ones_like(f).diag()` is (stages·N)² — 2.8·10^14 entries at
the benchmark state — updated by an outer product and factorised by `torch.linalg.solve` every iteration
(rk_common.py:438-459).  Here the same iteration runs matrix-free.  With J_0 = I every iterate has the form

    J_k = I + sum_{i<k} u_i s_i^T ,    u_i = (z_i - J_i s_i) / (s_i.s_i) = f_{i+1} / (s_i.s_i)      (J_i s_i = -f_i)

so  J_k s = -f_k  is solved with the Sherman–Morrison–Woodbury identity from a k x k system of DOT PRODUCTS
(k = iteration count, a handful):  s_k = -f_k + sum_i u_i [ (I + S^T U)^{-1} S^T f_k ]_i .  Per iteration that is two
`tdeq_multi_dot` launches (2k + 3 dots, one read-back), a `tdeq_weighted_sum` for s_k, one for K += s_k, and the stage
combines / residuals — O(k·N) bandwidth-bound work on the kernels of the explicit path instead of O(N²) memory and
O(N³) flops, which is what makes the methods usable at batch scale at all.

Parity: the iterates equal the reference's up to rounding (its LU solve vs. the closed form), and both stop at the
same test `||f||_2 < tol` (1e-6 fp32 / 1e-8 fp64), so solutions agree to about that tolerance, not bit for bit
(tests/test_implicit_golden.py).  Kept quirks: stages with alpha == 1 are evaluated at `nextafter(t1, -inf)` whether
or not `perturb` is set (rk_common.py:468-470 / :497-499); the convergence test sits at the top of the loop, so a
solve that converges only in its `max_iters`-th update still warns; a Python-float `dt` (event mode) is rounded
through the default dtype (`torch.tensor(dt)`, rk_common.py:421-422).

Gradients (plain `odeint` in grad mode).  The reference differentiates its unrolled Broyden iterations.  Here the
stage equations are solved without a graph and the solution K* is attached to the graph by the implicit function
theorem: the residual r(K*; y0, t, θ) is evaluated once more WITH a graph, and `_ImplicitCorrection` returns K* in
forward and  grad_r = -J^{-T} g  in backward, where J = dr/dK at K* and the linear system is solved matrix-free by
GMRES on vector-Jacobian products (autograd through `func` at the converged stage points).  That is the exact
gradient of the converged solution — O(1) graph memory per step instead of one graph per Broyden iteration.
`odeint_adjoint` works with every method as well (it needs only no-grad solves).
"""
from __future__ import annotations

import math
import warnings
from typing import Callable, List

import numpy as np
import torch

from .misc import Perturb
from .solvers import _NO_SHADOW, FixedGridODESolver
from .tableaus import IMPLICIT_TABLEAUS, ImplicitTableau

_DOT_TERMS = 14          # TDEQ_MAX_TERMS vectors per tdeq_multi_dot launch
_SUM_TERMS = 8           # TDEQ_MAX_SUM_TERMS per tdeq_weighted_sum launch


class _MatrixFreeBroyden:
    """Broyden's good method with J_0 = I on one flat unknown vector, never forming J (see the module docstring).
    `residual(K) -> f` must return a fresh contiguous tensor shaped like K."""

    def __init__(self, kernels, tol: float, max_iters: int):
        self.k = kernels
        self.tol = tol
        self.max_iters = max_iters
        self.n_iters = 0

    def _dots(self, g: torch.Tensor, xs: List[torch.Tensor]) -> torch.Tensor:
        return torch.cat([self.k.multi_dot(g, xs[lo:lo + _DOT_TERMS]) for lo in range(0, len(xs), _DOT_TERMS)])

    def _combo(self, vecs: List[torch.Tensor], ws: List[float]) -> torch.Tensor:
        """sum_j ws_j vecs_j (zero weights skipped), left to right."""
        pairs = [(v, w) for v, w in zip(vecs, ws) if w != 0.0]
        if not pairs:
            return torch.zeros_like(vecs[0])
        acc = None
        lo = 0
        while lo < len(pairs):
            room = _SUM_TERMS - (0 if acc is None else 1)
            part = pairs[lo:lo + room]
            out = torch.empty_like(vecs[0])
            xs = ([acc] if acc is not None else []) + [v for v, _ in part]
            wv = ([1.0] if acc is not None else []) + [w for _, w in part]
            self.k.weighted_sum(out, xs, wv)
            acc = out
            lo += room
        return acc

    def solve(self, K: torch.Tensor, residual: Callable[[torch.Tensor], torch.Tensor]):
        """Returns (K, converged)."""
        f = residual(K)
        fnorm2 = float(self.k.multi_dot(f, [f])[0])
        S: List[torch.Tensor] = []          # steps s_i
        F: List[torch.Tensor] = [f]         # residuals f_0 .. f_k   (u_i = rho_i f_{i+1})
        rho: List[float] = []               # 1 / (s_i . s_i)
        G = np.zeros((0, 0))                # G[i, j] = s_i . u_j
        d = np.zeros(0)                     # d_i = s_i . f_k
        converged = False
        self.n_iters = 0
        for _ in range(self.max_iters):
            if math.sqrt(fnorm2) < self.tol:            # NaN compares false, as in the reference
                converged = True
                break
            if not math.isfinite(fnorm2):
                break
            k = len(S)
            w = np.zeros(k + 1)
            w[k] = -1.0
            if k:
                try:
                    alpha = np.linalg.solve(np.eye(k) + G, d)
                except np.linalg.LinAlgError:           # the reference stops at a singular J (rk_common.py:444-447)
                    break
                if not np.all(np.isfinite(alpha)):
                    break
                w[1:] += alpha * np.asarray(rho)
            s = self._combo(F, w.tolist())
            K_new = torch.empty_like(K)
            self.k.weighted_sum(K_new, [K, s], [1.0, 1.0])
            K = K_new
            f = residual(K)
            self.n_iters += 1
            # one read-back: f.f, s_i.f (i <= k), s_k.s_k, s_k.f_{j+1} (j < k)
            vals = torch.cat([self._dots(f, [f] + S + [s]), self._dots(s, [s] + F[1:])]).tolist()
            fnorm2 = vals[0]
            d = np.asarray(vals[1:k + 2])
            ss = vals[k + 2]
            row = np.asarray(vals[k + 3:k + 3 + k]) * np.asarray(rho) if k else np.zeros(0)
            rho_k = 1.0 / ss if ss != 0.0 else float("inf")
            G_new = np.zeros((k + 1, k + 1))
            G_new[:k, :k] = G
            G_new[k, :k] = row
            G_new[:, k] = d * rho_k
            G = G_new
            rho.append(rho_k)
            S.append(s)
            F.append(f)
        return K, converged


def _gmres(matvec: Callable[[torch.Tensor], torch.Tensor], b: torch.Tensor, rtol: float, restart: int = 30,
           max_restarts: int = 10) -> torch.Tensor:
    """Restarted GMRES for A x = b with A given by `matvec`; vectors on the device, the small Hessenberg
    least-squares problem on the host.  Used only by the backward pass of `_ImplicitCorrection`."""
    x = torch.zeros_like(b)
    bnorm = float(b.norm())
    if bnorm == 0.0 or not math.isfinite(bnorm):
        return x
    for outer in range(max_restarts):
        r = b - matvec(x) if outer else b.clone()
        beta = float(r.norm())
        if beta <= rtol * bnorm:
            break
        V = [r / beta]
        H = np.zeros((restart + 1, restart))
        y, done, m = np.zeros(0), False, 0
        for j in range(restart):
            w = matvec(V[j])
            basis = torch.stack(V)
            h = basis @ w
            w = w - h @ basis
            h2 = basis @ w                      # second Gram–Schmidt pass
            w = w - h2 @ basis
            H[:j + 1, j] = (h + h2).tolist()
            hn = float(w.norm())
            H[j + 1, j] = hn
            m = j + 1
            e1 = np.zeros(m + 1)
            e1[0] = beta
            y = np.linalg.lstsq(H[:m + 1, :m], e1, rcond=None)[0]
            resid = float(np.linalg.norm(H[:m + 1, :m] @ y - e1))
            if resid <= rtol * bnorm or hn <= 1e-300:
                done = True
                break
            V.append(w / hn)
        x = x + torch.as_tensor(y, dtype=b.dtype, device=b.device) @ torch.stack(V[:m])
        if done:
            break
    return x


class _ImplicitCorrection(torch.autograd.Function):
    """forward: K* (the no-graph solution of r(K) = 0); backward: grad_r = -J^{-T} g with J = dr/dK at K* (implicit
    function theorem), J^T v supplied by `vjp`."""

    @staticmethod
    def forward(ctx, r, K_star, vjp, rtol):
        ctx.vjp, ctx.rtol = vjp, rtol
        return K_star.clone()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        lam = _gmres(ctx.vjp, g.contiguous(), ctx.rtol)
        return -lam, None, None, None


class FixedGridImplicitRKSolver(FixedGridODESolver):
    """FIRK (all stages coupled, rk_common.py:378-479) and DIRK (stage by stage, :482-558) drivers."""
    tableau: ImplicitTableau

    def __init__(self, func, y0, max_iters=100, residual_tol=None, residual_norm="l2", **kwargs):
        super().__init__(func, y0, **kwargs)
        self.max_iters = max_iters
        self.order = self.tableau.order
        T = func.np_dtype
        # The reference stops at an ABSOLUTE 2-norm of the stacked residual (1e-6 fp32 / 1e-8 fp64, rk_common.py:424-428).
        # That bound does not scale: at 10^7 unknowns the rounding floor of an fp32 residual (eps * |K| * sqrt(n))
        # is above 1e-6 and every step would run all `max_iters` iterations.  Two opt-in extensions for batch-scale
        # states: `residual_tol` replaces the bound, `residual_norm="rms"` tests ||f||_2 / sqrt(n) instead.
        self._tol = float(residual_tol) if residual_tol is not None else (1e-6 if y0.dtype == torch.float32 else 1e-8)
        if residual_norm not in ("l2", "rms"):
            raise ValueError("residual_norm must be 'l2' (the reference's test) or 'rms'")
        self._rms = residual_norm == "rms"
        tab = self.tableau
        self._alpha = [T(a) for a in tab.alpha]                       # tableau cast to the state dtype (:412-415)
        self._beta = [[float(T(b)) for b in row] for row in tab.beta]
        self._n = self.layout.total
        self._stride = -(-self._n // 4) * 4                           # stage stride of the stacked unknown: 16 B aligned

    # -- stage time / perturbation / skip rule (rk_common.py:462-479, :497-507) -------------------------
    def _stage(self, i: int, t0T, dtT, t1T):
        """(time, perturb) of stage i, or None when the stage keeps the stored slope f0."""
        T = type(t0T)
        a = self._alpha[i]
        if a == T(1):
            return t1T, Perturb.PREV
        if a == T(0):
            if not all(b != 0.0 for b in self._beta[i]):
                return None
            return t0T, Perturb.NONE
        return T(t0T + T(a * dtT)), Perturb.NONE

    def _step(self, t0, dt, t1, y0, y1_out, sh):
        func, kern, ops = self.func, self.kernels, self.ops
        T = func.np_dtype
        f0 = func.eval(t0, y0, self._first_perturb(), shadow=sh.time(0.0))
        graph = torch.is_grad_enabled() and (y0.requires_grad or f0.requires_grad or sh is not _NO_SHADOW)
        if isinstance(dt, float):
            dt = torch.tensor(dt).item()       # `torch.tensor(dt)`: a Python float passes through the default dtype
        t0T, dtT, t1T = T(t0), T(dt), T(t1)
        dts = float(dtT) * func.sign
        n_st = len(self._alpha)
        stages = [self._stage(i, t0T, dtT, t1T) for i in range(n_st)]
        n, stride = self._n, self._stride
        n_unknown = n * (1 if self.tableau.diagonal else n_st)
        solver = _MatrixFreeBroyden(kern, self._tol * (math.sqrt(max(n_unknown, 1)) if self._rms else 1.0),
                                    self.max_iters)
        y0_d, f0_d = y0.detach(), f0.detach()
        gm_tol = 1e-6 if y0.dtype == torch.float32 else 1e-11

        def stage_input(ks: List[torch.Tensor], row: List[float]) -> torch.Tensor:
            nz = [(k, b) for k, b in zip(ks, row) if b != 0.0]
            out = torch.empty_like(y0_d)
            kern.stage_combine(out, y0_d, [k for k, _ in nz], [b for _, b in nz], dts)
            return out

        def stage_residual(i, st, ks, y_base, f_base, with_time):
            """r_i = K_i - f(t_i, y_base + dt sum_j beta_ij K_j) through the recording front end (grad mode)."""
            if st is None:                                   # stored slope: K_i = f0
                return ops.weighted_sum([ks[i], f_base], [1.0, -1.0])
            row = self._beta[i]
            nz = [(k, b) for k, b in zip(ks, row) if b != 0.0]
            yi = ops.combine(y_base, [k for k, _ in nz], [b for _, b in nz], dts, sh.dt_signed() if with_time else None)
            a = 1.0 if st[1] is Perturb.PREV else float(self._alpha[i])
            fi = func.eval(st[0], yi, st[1], shadow=sh.time(a) if with_time else None)
            return ops.weighted_sum([ks[i], fi], [1.0, -1.0])

        def warn_if(not_converged: bool) -> None:
            if not_converged:
                warnings.warn("Functional iteration did not converge. Solution may be incorrect.")

        if not self.tableau.diagonal:
            K = torch.zeros(n_st * stride, dtype=y0.dtype, device=y0.device)
            for i in range(n_st):
                K[i * stride:i * stride + n].copy_(f0_d)
            split = lambda Kf: [Kf[j * stride:j * stride + n] for j in range(n_st)]

            def residual(Kf: torch.Tensor) -> torch.Tensor:
                ks = split(Kf)
                res = torch.empty_like(Kf)
                for i, st in enumerate(stages):
                    if st is None:                       # stored slope: this block of the residual is identically zero
                        res[i * stride:(i + 1) * stride].zero_()
                        continue
                    fi = func.eval(st[0], stage_input(ks, self._beta[i]), st[1])
                    kern.weighted_sum(res[i * stride:i * stride + n], [ks[i], fi], [1.0, -1.0])
                    if stride > n:
                        res[i * stride + n:(i + 1) * stride].zero_()      # alignment padding of the stacked unknown
                return res

            with torch.no_grad():
                K, converged = solver.solve(K, residual)
            warn_if(not converged)
            if graph:
                pad = [torch.zeros(stride - n, dtype=y0.dtype, device=y0.device)] if stride > n else []

                def residual_graph(Kf, y_base, f_base, with_time):
                    ks = split(Kf)
                    pieces = []
                    for i, st in enumerate(stages):
                        pieces.append(stage_residual(i, st, ks, y_base, f_base, with_time))
                        pieces.extend(pad)
                    return torch.cat(pieces)

                with torch.enable_grad():
                    K_leaf = K.detach().requires_grad_(True)
                    r_leaf = residual_graph(K_leaf, y0_d, f0_d, False)
                    r = residual_graph(K.detach(), y0, f0, True)
                vjp = lambda v: torch.autograd.grad(r_leaf, K_leaf, v, retain_graph=True)[0]
                K = _ImplicitCorrection.apply(r, K.detach(), vjp, gm_tol) if r.requires_grad else K
            ks = split(K)
            if graph:       # stored-slope stages carry f0's graph directly as well (their residual row is K_i - f0)
                ks = [k.contiguous() for k in ks]
        else:
            ks = [f0 if graph else f0_d] * n_st
            for i, st in enumerate(stages):
                if st is None:
                    continue
                prev_d = [k.detach() for k in ks[:i]]

                def residual(ki: torch.Tensor, i=i, st=st, prev_d=prev_d) -> torch.Tensor:
                    fi = func.eval(st[0], stage_input(prev_d + [ki], self._beta[i]), st[1])
                    res = torch.empty_like(ki)
                    kern.weighted_sum(res, [ki, fi], [1.0, -1.0])
                    return res

                with torch.no_grad():
                    ki, converged = solver.solve(ks[i].detach(), residual)
                warn_if(not converged)
                if graph:
                    with torch.enable_grad():
                        k_leaf = ki.detach().requires_grad_(True)
                        r_leaf = stage_residual(i, st, prev_d + [k_leaf], y0_d, f0_d, False)
                        r = stage_residual(i, st, list(ks[:i]) + [ki.detach()], y0, f0, True)
                    vjp = lambda v, r_leaf=r_leaf, k_leaf=k_leaf: torch.autograd.grad(r_leaf, k_leaf, v,
                                                                                      retain_graph=True)[0]
                    if r.requires_grad:
                        ki = _ImplicitCorrection.apply(r, ki.detach(), vjp, gm_tol)
                ks = ks[:i] + [ki] + ks[i + 1:]
        c_sol = [float(T(c)) for c in self.tableau.c_sol]
        nz = [(k, c) for k, c in zip(ks, c_sol) if c != 0.0]
        y1 = ops.combine(y0, [k for k, _ in nz], [c for _, c in nz], dts, sh.dt_signed(),
                         out=None if graph else y1_out)
        return y1, f0


def _make(name: str):
    tab = IMPLICIT_TABLEAUS[name]
    return type(name, (FixedGridImplicitRKSolver,), {"tableau": tab, "order": tab.order,
                                                     "__doc__": f"`{name}` (fixed_grid_implicit.py)."})


ImplicitEuler = _make("implicit_euler")
ImplicitMidpoint = _make("implicit_midpoint")
Trapezoid = _make("trapezoid")
RadauIIA3 = _make("radauIIA3")
GaussLegendre4 = _make("gl4")
RadauIIA5 = _make("radauIIA5")
GaussLegendre6 = _make("gl6")
SDIRK2 = _make("sdirk2")
TRBDF2 = _make("trbdf2")
