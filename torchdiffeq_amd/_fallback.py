"""Host path for states the HIP kernels do not take: tensors that are NOT on a ROCm device, and complex states.

The reference runs wherever its tensors live (torchdiffeq/_impl/odeint.py:49-108 — BASELINE.json's configs[0] is
"rk4 ... fp32 on CPU (plumbing, no GPU)" — and misc.py:185 / rk_common.py:61 use `abs().dtype` so that complex states
work).  `HostKernels` gives the solver drivers the same kernel interface as `_native.HipKernels`, written with plain
torch ops on the state's own device, so that such a call is a drop-in too instead of an error.

Selection is by the STATE alone (`_native.get_kernels`): a real fp32 / fp64 state on a ROCm device always takes the
HIP kernels and fails loudly if libtdeq_hip.so is missing — it never lands here.  This module never imports `oracle/`
(tests/test_abi.py::test_product_never_imports_oracle); the first use warns once (`HostPathWarning`).

Arithmetic (r04): the REFERENCE's, literally.  Where the HIP kernels have to pick an order ATen leaves open — the sum
over a tableau row, the accumulation of a norm — this path does not pick: it hands ATen the same expression the reference
evaluates.  A row sum is `torch.sum` over a dense [N, row length] product tensor with the products at their tableau
positions (`_rowsum`: ATen adds ≤ 7 columns as four interleaved partial sums, more through vector lanes — the result
depends on the positions, docs/LAB_NOTEBOOK.md §8); a norm is `x.abs().pow(2).mean().sqrt()` in the state's type per component
(`HostPlan.rms0 / rms1`, misc.py:22-33), the fp64 sums the kernel interface reports are kept beside it.  Everything else
follows the operation order of interp.py:17-21,42-47 and rk_common.py:110-157 with coefficients fl_T(fl_T(coef) *
fl_T(dt)) (rk_common.py:79,89,201-205), and the host scalars round like 0-dim tensors (`_scalars.py`).  Consequence: on
the CPU every explicit method — adaptive and fixed-grid, fp32 / fp64 / complex / bf16, adjoint included — reproduces the
reference BIT FOR BIT (tests/test_hostpath.py, tests/test_brow_golden.py, `tools/fuzz_vs_reference.py` with
TDEQ_FUZZ_BACKEND=host); the solver drivers switch off the fused error split and the carried partial sums here (both
re-associate a row).  T = the REAL dtype of the state (`y0.abs().dtype`), also for complex states.
"""
from __future__ import annotations

import math
import warnings
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._scalars import operand, scalar_type


class HostPathWarning(UserWarning):
    """torchdiffeq_amd is integrating a state with torch ops instead of the MI355X HIP kernels."""


_warned = False


def warn_once(reason: str) -> None:
    global _warned
    if not _warned:
        _warned = True
        warnings.warn(f"torchdiffeq_amd: {reason}; this solve runs on the package's torch-op host path, not on the "
                      "MI355X HIP kernels (move a real fp32 / fp64 state to a ROCm device for those)", HostPathWarning,
                      stacklevel=3)


def real_np_dtype(dtype: torch.dtype):
    """Host scalar type of `y0.abs().dtype` (misc.py:185, rk_common.py:61)."""
    return scalar_type(dtype)


class HostPlan:
    """Segment table and result words of the norm operations for one state layout (`_native.NormPlan`'s role)."""

    def __init__(self, segments: Sequence[Tuple[int, int, float, float]], total: int, chunk: int):
        self.chunk = chunk
        self.n_seg = len(segments)
        self.numels = [int(s[1]) for s in segments]
        self.n_chunks = max(1, -(-total // chunk))
        self.segs = [(int(off), int(n), float(rt), float(at)) for off, n, rt, at in segments]
        self.segs_dev = None
        self.pinned = False
        self.expect = ()
        self.sums0 = [0.0] * self.n_seg
        self.sums1 = [0.0] * self.n_seg
        self.bad = [0.0] * self.n_seg
        # the fp64 sums of squares are consumed by lock-step sharding only (dist_sync all-reduces them); an unsharded solve
        # on this path reads rms0 / rms1, so the extra fp64 reduction per norm is skipped unless asked for (advisor r05)
        self.want_sums = False
        # the reference's own norm of the same quotients, per segment: sqrt(mean(|x|^2)) evaluated by ATen in the state's
        # type (what the solver drivers use on this path — `literal_norms`)
        self.rms0 = [0.0] * self.n_seg
        self.rms1 = [0.0] * self.n_seg
        # |x| of a one-element segment, not squared: the adjoint norms take their time component as `t.abs()`
        # (adjoint.py:250, 273) — the rms of one element is the same number unless its square leaves the type's range
        self.abs0 = [math.nan] * self.n_seg
        self.abs1 = [math.nan] * self.n_seg
        self.uniform_tol = len({(s[2], s[3]) for s in self.segs}) <= 1
        self._join = None

    def join_index(self, device) -> torch.Tensor:
        """Element indices of the segments, back to back — the reference's UNPADDED flat state (misc.py:206-209)."""
        if self._join is None or self._join.device != device:
            self._join = torch.cat([torch.arange(off, off + n, device=device) for off, n, _, _ in self.segs]) \
                if self.segs else torch.zeros(0, dtype=torch.long, device=device)
        return self._join


def _nonfinite(*xs: torch.Tensor) -> float:
    ok = torch.isfinite(xs[0])
    for x in xs[1:]:
        ok = ok & torch.isfinite(x)
    return float((~ok).sum())


def real_dot(g: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Re sum conj(g) x, accumulated in double precision (differentiable torch ops)."""
    if g.is_complex() or x.is_complex():
        return (g.to(torch.complex128).conj() * x.to(torch.complex128)).real.sum()
    return (g.double() * x.double()).sum()


def _no_grad_methods(cls):
    """The kernel interface computes VALUES (autograd is autodiff._LinearOp's business): no graph is recorded here."""
    for name, fn in list(vars(cls).items()):
        if callable(fn) and not name.startswith("_") and not isinstance(fn, (staticmethod, classmethod)):
            setattr(cls, name, torch.no_grad()(fn))
    return cls


@_no_grad_methods
class HostKernels:
    """`_native.HipKernels`' tensor-level interface in torch ops (elementwise in the state dtype, fp64 norm sums).
    The device-resident controller, the look-ahead stage and hipGraph capture are properties of the HIP path and do
    not exist here (`device_controller = False`): the host loop of solvers.py drives every step."""

    name = "host"
    device_controller = False
    literal_row_sums = True      # no fused error split / carried partial sums on this path (solvers.py)
    literal_norms = True         # HostPlan.rms0 / rms1 hold the reference's norm values

    # -- helpers -----------------------------------------------------------------------------------------
    @staticmethod
    def _T(x: torch.Tensor):
        return real_np_dtype(x.dtype)

    @classmethod
    def _coefs(cls, like: torch.Tensor, coefs: Sequence[float], dt: float) -> List[float]:
        T = cls._T(like)
        dtT = T(dt)
        return [float(T(T(c) * dtT)) for c in coefs]

    @staticmethod
    def _lsum(ks: Sequence[torch.Tensor], cs: Sequence[float], start: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(start + c0 k0) + c1 k1 ... left to right; without `start` the first product starts the sum."""
        acc = start
        for k, c in zip(ks, cs):
            p = k * c
            acc = p if acc is None else acc + p
        return acc

    @staticmethod
    def _sumsq(r: torch.Tensor) -> float:
        """Sum of |r|^2 in fp64 — what the norm kernels hand back per segment (the host forms sqrt(sum / n))."""
        if r.is_complex():          # re^2 + im^2 in double, like the complex norm kernels (not the square of a rounded modulus)
            v = torch.view_as_real(r).double()
            return float((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]).sum())
        return float(r.double().pow(2).sum())

    @staticmethod
    def _rowsum(ks, cs, start=None, row=None):
        """A tableau row's sum — the reference's `torch.sum(k[..., :n] * c, dim=-1)` (rk_common.py:79,89,366) — as opposed
        to a chain of elementwise additions (`_lsum`: rk_common.py:110-157, solvers.py:166-181, fixed_adams.py).  `row`
        = the row's `tableaus.RowCoefs` (positions `idx` of the non-zero weights in a dense row of `width` columns): the
        products go to their columns of a zero-filled [N, width] tensor (a zero weight contributes +0, but its position
        decides which of ATen's partial sums the others land in) and ATen sums it — for every dtype its own way (bf16 /
        fp16: products rounded to the state type, accumulated in fp32, rounded once).  Without `row` (user-supplied
        weights, continued partial sums): the chain."""
        width = getattr(row, "width", None)
        if start is not None or width is None or len(ks) == 0:
            return HostKernels._lsum(ks, cs, start)
        prod = ks[0].new_zeros((ks[0].numel(), width))
        for k, c, j in zip(ks, cs, row.idx):
            torch.mul(k.reshape(-1), c, out=prod[:, j])
        return prod.sum(dim=-1).view_as(ks[0])

    @staticmethod
    def _rms(r: torch.Tensor) -> float:
        """misc.py:22-23, literally (NaN for an empty component, as there)."""
        return float(r.abs().pow(2).mean().sqrt()) if r.numel() else float("nan")

    def make_plan(self, segments, total, chunk, device) -> HostPlan:
        plan = HostPlan(segments, total, chunk)
        # who reads the fp64 sums: the variant whose norms ARE those sums (KernelOrderHostKernels), and lock-step sharding
        # (the solver sets the flag then); the literal path reads rms0 / rms1
        plan.want_sums = not self.literal_norms
        return plan

    # -- stage combines ------------------------------------------------------------------------------------
    def stage_combine(self, out, y0, ks, coefs, dt: float) -> None:
        torch.add(y0, self._rowsum(ks, self._coefs(y0, coefs, dt), None, coefs), out=out)

    def stage_combine_fill(self, out, y0, ks, coefs, dt: float, fill_dst, fill_vals) -> None:
        self.stage_combine(out, y0, ks, coefs, dt)
        self.fill_scalars(fill_dst, fill_vals)

    def stage_combine_err(self, out, err_out, y0, ks, coefs, err_coefs, dt: float) -> None:
        self.stage_combine(out, y0, ks, coefs, dt)
        err_out.copy_(self._rowsum(ks, self._coefs(y0, err_coefs, dt)))

    def stage_combine_multi(self, outs, rows, y0, acc_in, ks, dt: float, events=None) -> None:
        for o, (out, (coefs, mask, add_y0)) in enumerate(zip(outs, rows)):
            cs = self._coefs(y0, coefs, dt)
            take = [j for j in range(len(ks)) if (mask >> j) & 1]
            s = self._rowsum([ks[j] for j in take], [cs[j] for j in take], acc_in if o == 0 else None)
            if add_y0:
                torch.add(y0, s, out=out)
            else:
                out.copy_(s)

    # -- norms -----------------------------------------------------------------------------------------------
    @staticmethod
    def _joint(plan: HostPlan, like: torch.Tensor) -> bool:
        """Complex states of several segments: |z| and z / real are not single IEEE operations, and ATen's vectorised
        loop and its scalar tail round them differently in the last bit (~1 element in 1e3) — which of the two an
        element gets depends on its POSITION in the tensor the operation runs on.  The reference runs them on the
        unpadded concatenation of all components (rk_common.py:22-27, misc.py:50-66), so the quotients are formed on
        that same tensor here and sliced afterwards (real types: every operation is exactly rounded, position-free)."""
        return like.is_complex() and plan.n_seg > 1 and plan.uniform_tol

    def _error_sums(self, plan: HostPlan, e, y0, y1, scaled_out) -> None:
        T = self._T(y0)
        # misc.py:81 `torch.max(y0.abs(), y1.abs())` propagates a NaN of the trial state into the tolerance and the ratio
        # (the step is then rejected with a NaN step size); the kernels' fmax ignores it and leaves the verdict to their
        # non-finite census (docs/LAB_NOTEBOOK.md §8) — same exception, but not the same number of trial steps before it
        larger = torch.maximum if self.literal_norms else torch.fmax
        joint = None
        if self._joint(plan, y0):
            idx = plan.join_index(y0.device)
            _, _, rtol, atol = plan.segs[0]
            joint = e[idx] / (larger(y0[idx].abs(), y1[idx].abs()) * float(T(rtol)) + float(T(atol)))
            lo = 0
        for s, (off, n, rtol, atol) in enumerate(plan.segs):
            sl = slice(off, off + n)
            if joint is not None:
                r = joint[lo:lo + n]
                lo += n
            else:
                tol = larger(y0[sl].abs(), y1[sl].abs()) * float(T(rtol)) + float(T(atol))
                r = e[sl] / tol
            plan.rms0[s] = rms = self._rms(r)
            plan.abs0[s] = float(r.abs()) if n == 1 else math.nan
            plan.sums0[s] = self._sumsq(r) if plan.want_sums else math.nan    # the fp64 sum itself: lock-step sharding (dist_sync) all-reduces it
            plan.bad[s] = _nonfinite(y0[sl], y1[sl])
            if scaled_out is not None:
                scaled_out[sl] = r
        if scaled_out is not None and plan.n_seg > 1:          # padding of a segmented layout: zeros
            hi = 0
            for off, n, _, _ in plan.segs:
                if off > hi:
                    scaled_out[hi:off].zero_()
                hi = off + n
            scaled_out[hi:].zero_()

    def error_norm(self, plan, y0, y1, ks, coefs, dt: float, scaled_out=None) -> None:
        self._error_sums(plan, self._rowsum(ks, self._coefs(y0, coefs, dt), None, coefs), y0, y1, scaled_out)

    def error_norm_partial(self, plan, err_partial, y0, y1, ks, coefs, dt: float) -> None:
        e = self._rowsum(ks, self._coefs(y0, coefs, dt), err_partial) if len(ks) else err_partial
        self._error_sums(plan, e, y0, y1, None)

    def error_scaled(self, plan, out, y0, y1, ks, coefs, dt: float) -> None:
        self.error_norm(plan, y0, y1, ks, coefs, dt, scaled_out=out)

    def _init_quotients(self, plan, mode, a, b, yscale):
        T = self._T(yscale)
        if self._joint(plan, yscale):           # see _joint: the reference's tensor, sliced afterwards
            idx = plan.join_index(yscale.device)
            _, _, rtol, atol = plan.segs[0]
            scale = yscale[idx].abs() * operand(T, rtol) + float(T(atol))
            q0 = (a[idx] / scale) if mode == 0 else ((a[idx] - b[idx]) / scale)
            q1 = (b[idx] / scale) if mode == 0 else None
            lo = 0
            for s, (off, n, _, _) in enumerate(plan.segs):
                yield s, slice(off, off + n), q0[lo:lo + n], None if q1 is None else q1[lo:lo + n]
                lo += n
            return
        for s, (off, n, rtol, atol) in enumerate(plan.segs):
            sl = slice(off, off + n)
            scale = yscale[sl].abs() * operand(T, rtol) + float(T(atol))       # misc.py:50: |y0| * rtol, then atol + ...
            if mode == 0:
                yield s, sl, a[sl] / scale, b[sl] / scale
            else:
                yield s, sl, (a[sl] - b[sl]) / scale, None

    def init_norms(self, plan, mode: int, a, b, yscale) -> None:
        for s, sl, q0, q1 in self._init_quotients(plan, mode, a, b, yscale):
            n = q0.numel()
            plan.rms0[s] = rms = self._rms(q0)
            plan.abs0[s] = float(q0.abs()) if n == 1 else math.nan
            plan.sums0[s] = self._sumsq(q0) if plan.want_sums else math.nan
            if q1 is not None:
                plan.rms1[s] = rms = self._rms(q1)
                plan.abs1[s] = float(q1.abs()) if n == 1 else math.nan
                plan.sums1[s] = self._sumsq(q1) if plan.want_sums else math.nan
            plan.bad[s] = _nonfinite(yscale[sl])

    def init_scaled(self, plan, mode: int, a, b, yscale, out0, out1=None) -> None:
        if plan.n_seg > 1:
            out0.zero_()
            if out1 is not None:
                out1.zero_()
        for s, sl, q0, q1 in self._init_quotients(plan, mode, a, b, yscale):
            out0[sl] = q0
            if q1 is not None:
                out1[sl] = q1

    def read_norms(self, plan):
        return list(plan.sums0), list(plan.sums1), list(plan.bad)

    # -- dense output (rk_common.py:363-369, interp.py:1-48) -----------------------------------------------------
    def _quartic(self, y0, y1, f0, f1, ks, coefs, dt: float):
        T = self._T(y0)
        dtT = float(T(dt))
        ymid = y0 + self._rowsum(ks, self._coefs(y0, coefs, dt), None, coefs)
        two_dt = float(T(T(2) * T(dt)))
        qa = (two_dt * (f1 - f0) - 8 * (y1 + y0)) + 16 * ymid
        qb = ((dtT * (5 * f0 - 3 * f1) + 18 * y0) + 14 * y1) - 32 * ymid
        qc = ((dtT * (f1 - 4 * f0) - 11 * y0) - 5 * y1) + 16 * ymid
        qd = dtT * f0
        return y0, qd, qc, qb, qa

    def _poly(self, q, x: float, like):
        T = self._T(like)
        xT = T(x)
        qe, qd, qc, qb, qa = q
        total = qe + float(xT) * qd
        xp = T(xT * xT)
        total = total + float(xp) * qc
        xp = T(xp * xT)
        total = total + float(xp) * qb
        xp = T(xp * xT)
        return total + float(xp) * qa

    def dense_eval(self, out, y0, y1, f0, f1, ks, coefs, dt: float, x: float) -> None:
        out.copy_(self._poly(self._quartic(y0, y1, f0, f1, ks, coefs, dt), x, y0))

    def dense_eval_multi(self, out_rows, y0, y1, f0, f1, ks, coefs, dt: float, xs: Sequence[float]) -> None:
        q = self._quartic(y0, y1, f0, f1, ks, coefs, dt)
        for row, x in zip(out_rows, xs):
            row.copy_(self._poly(q, x, y0))

    def interp_fit(self, coeffs, y0, y1, f0, f1, ks, coefs, dt: float) -> None:
        for plane, q in zip(coeffs, self._quartic(y0, y1, f0, f1, ks, coefs, dt)):
            plane.copy_(q)

    # -- fixed-grid steps (rk_common.py:110-157, solvers.py:166-181) ------------------------------------------------
    def rk4_stage(self, stage: int, out, y0, k1, k2, k3, k4, dt: float) -> None:
        # rk_common.py:110-118 with dt a 0-dim tensor: `dt * k` rounds dt to the state type first, `k * dt` and
        # `... * _one_third` take the scalar as ATen takes a second operand (`_scalars.operand`: the state type for
        # fp32 / fp64 — the same number —, fp32 for the 16-bit types)
        T = self._T(y0)
        dt_first, dt_second, third = float(T(dt)), operand(T, dt), operand(T, 1.0 / 3.0)
        if stage == 1:
            r = y0 + (dt_first * k1) * third
        elif stage == 2:
            r = y0 + dt_first * (k2 - k1 * third)
        elif stage == 3:
            r = y0 + dt_first * ((k1 - k2) + k3)
        else:
            r = y0 + (((k1 + 3 * (k2 + k3)) + k4) * dt_second) * 0.125
        out.copy_(r)

    def scaled_add(self, out, y0, k, scalar: float) -> None:
        """y0 + k * s with the 0-dim scalar as the SECOND operand (fixed_grid.py:18, `y0 + f0 * half_dt`): the state type
        for fp32 / fp64 — where it equals the generic stage combine —, fp32 for the 16-bit types (`_scalars.operand`)."""
        out.copy_(y0 + k * operand(self._T(y0), scalar))

    def lerp(self, out, y0, y1, slope: float) -> None:
        out.copy_(y0 + float(self._T(y0)(slope)) * (y1 - y0))

    def fixed_stage(self, mode: int, out, y0, ks, ws, dt: float) -> None:
        # rk_common.py:121-157: `dt * k1 * w` / `dt * (k1 * w1 + k2 * w2 ...)` — dt a 0-dim tensor as FIRST operand
        # (rounded to the state's type), the weights Python numbers as SECOND operands (`operand`: at fp32 next to a
        # 16-bit tensor, in the state's type otherwise)
        T = self._T(y0)
        dtT = float(T(dt))
        if mode == 1:
            out.copy_(y0 + (ks[0] * dtT) * operand(T, ws[0]))
            return
        out.copy_(y0 + self._lsum(ks, [operand(T, w) for w in ws]) * dtT)

    def weighted_sum(self, out, xs, ws) -> None:
        T = self._T(out)
        out.copy_(self._lsum(xs, [float(T(w)) for w in ws]))

    def scale_many(self, outs, g, ws) -> None:
        T = self._T(g)
        for o, w in zip(outs, ws):
            torch.mul(g, float(T(w)), out=o)

    def multi_dot(self, g, xs) -> torch.Tensor:
        """fp64 [len(xs)] of Re <g, x_m> (= sum g x_m for real states): the derivative of a real loss wrt a real
        scalar s of out = sum_m w_m(s) x_m, given g = dL/d conj(out) as autograd hands it over."""
        return torch.stack([real_dot(g, x) for x in xs])

    # -- Adams–Bashforth(–Moulton) (fixed_adams.py:160-223) ------------------------------------------------------
    def adams_predict(self, y_out, y0, hist, cb, cm=None, dt: float = 0.0, dy_out=None, delta_out=None) -> None:
        T = self._T(y0)
        dy = None
        for f, c in zip(hist, cb):
            p = float(T(c)) * f
            dy = p if dy is None else dy + p
        torch.add(y0, dy, out=y_out)
        if dy_out is not None:
            sm = None
            for f, c in zip(hist, cm):
                p = float(T(c)) * f
                sm = p if sm is None else sm + p
            dy_out.copy_(dy)
            delta_out.copy_(float(T(dt)) * sm)

    def adams_correct(self, plan, dy_out, dy_old, y_out=None, f=None, delta=None, y0=None, c: float = 0.0,
                      compute: bool = True) -> None:
        T = self._T(dy_out)
        if compute:
            d = float(T(c)) * f + delta
            dy_out.copy_(d)
            torch.add(y0, d, out=y_out)
        for s, (off, n, rtol, atol) in enumerate(plan.segs):
            sl = slice(off, off + n)
            d0, d1 = dy_old[sl], dy_out[sl]
            tol = torch.fmax(d0.abs(), d1.abs()) * float(T(rtol)) + float(T(atol))
            r = (d0 - d1).abs() / tol
            plan.sums0[s] = float((~(r < 1)).sum())
            plan.bad[s] = _nonfinite(d1)

    # -- packing / scalars -------------------------------------------------------------------------------------
    def pack_segments(self, out, srcs, chunk_starts, numels, scales, chunk: int) -> None:
        out.zero_()
        for t, cs, n, sc in zip(srcs, chunk_starts, numels, scales):
            if t is None or n == 0:
                continue
            dst = out[cs * chunk:cs * chunk + n]
            dst.copy_(t.reshape(-1) if sc == 1.0 else t.reshape(-1) * sc)

    def fill_scalars(self, dst, vals: Sequence[float]) -> None:
        dst.copy_(torch.tensor(list(vals), dtype=torch.float64).to(dst.dtype))

    # `arm_readback` / `read_ctrl` / `stage_combine_sel` / `stage_combine_dev` / `step_commit` / `step_controller`:
    # deliberately absent (see the class docstring); solvers.py checks `device_controller`.


@_no_grad_methods
class LowPrecisionHostKernels(HostKernels):
    """bfloat16 / float16 states: the reference integrates them in their own precision (misc.py:185-187,
    rk_common.py:61-65 — every time-like scalar is cast to `y0.abs().dtype`), and so does this backend, on whatever
    device the state lives.  The arithmetic is HostKernels' — ATen itself rounds a reduced-precision `torch.sum`
    (products in the state type, fp32 accumulation, one rounding) and `mean()` its own way, and `_scalars.operand` /
    `BFloat16Scalar` / `Float16Scalar` reproduce how it takes Python numbers and 0-dim partners next to a 16-bit tensor
    (tools/lowfloat_semantics.py).  A class of its own only so that the selection (`_native.get_kernels`) and the
    warning say what runs: no HIP kernels exist for these element types (docs/LAB_NOTEBOOK.md §10)."""

    name = "host-low"


@_no_grad_methods
class KernelOrderHostKernels(HostKernels):
    """The HIP kernels' arithmetic restated in torch ops: tableau rows summed LEFT TO RIGHT over the non-zero entries
    (the order the kernels and the C oracle use, and what makes the fused error split and the carried partial sums
    possible), norms from the fp64 sums.  Not selected for any state — it exists so that the GPU tests and
    tools/fuzz_complex_gpu.py can compare a HIP solve with the same algorithm evaluated by ATen on the same device
    (test infrastructure; r03's host path did this for every state)."""

    name = "host-kernel-order"
    literal_row_sums = False
    literal_norms = False

    @staticmethod
    def _rowsum(ks, cs, start=None, row=None):
        return HostKernels._lsum(ks, cs, start)


@_no_grad_methods
class KernelOrderLowHostKernels(LowPrecisionHostKernels):
    """The 16-bit HIP kernels' arithmetic (csrc/tdeq_kernels_lp.hpp) restated in torch ops: a tableau row summed over its
    non-zero weights LEFT TO RIGHT with the rounded products accumulated in float32 and ONE rounding of the sum — the only
    place where `LowPrecisionHostKernels` hands ATen a choice the kernels make themselves.  Not selected for any state:
    test infrastructure for same-device comparisons with `_lowp.LowPrecisionHipKernels` (tests/test_lowp_oracle.py,
    tools/fuzz_complex_gpu.py with FUZZ_LOW=1)."""

    name = "host-low-kernel-order"
    literal_row_sums = False
    split_row_sums = False       # like the 16-bit kernels: a row is rounded once, never continued by a second launch

    @staticmethod
    def _rms(r: torch.Tensor) -> float:
        """sqrt(mean(|x|^2)) as the 16-bit path forms it from the kernels' words: squares rounded to the type (ATen's
        `pow(2)`), their sum in fp64 (ATen: a float32 cascade — the one difference, far below the type's rounding), float32
        sum / n rounded once, sqrt rounded (`_lowp.LowPrecisionHipKernels.read_norms`)."""
        n = r.numel()
        if n == 0:
            return float("nan")
        T = real_np_dtype(r.dtype)
        total = float(r.abs().pow(2).double().sum())
        with np.errstate(all="ignore"):
            return float(T(np.float32(total) / np.float32(n)) ** 0.5)

    @staticmethod
    def _rowsum(ks, cs, start=None, row=None):
        acc = None if start is None else start.float()
        for k, c in zip(ks, cs):
            p = (k * c).float()
            acc = p if acc is None else acc + p
        return acc.to(ks[0].dtype)
