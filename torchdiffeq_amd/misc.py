"""Host-side input normalisation for `odeint` / `odeint_adjoint`.

Mirrors the observable behaviour of the reference's `_check_inputs` (torchdiffeq/_impl/misc.py:200-345)
— tuple states, decreasing time, `perturb`, callbacks, default norms, the same exceptions — but is
organised for the HIP path instead of a stack of `nn.Module` wrappers:

  * a (tuple) state becomes ONE flat buffer with chunk-aligned segments (`StateLayout`), so a single
    kernel launch covers the whole state and the norm kernels see the segment table;
  * decreasing time is a sign carried next to `dt` (bit-identical to multiplying every func output by
    -1, see include/tdeq_hip.h) instead of an extra full-state multiply per evaluation;
  * time perturbation (`nextafter`) and the cast of `t` to the state dtype are host scalar arithmetic.
"""
from __future__ import annotations

import functools
import math
import os
import warnings
from enum import Enum
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native
from ._scalars import nextafter, real_dtype, scalar_type
from .autodiff import stitch

SUPPORTED_STATE_DTYPES = (torch.float32, torch.float64, torch.complex64, torch.complex128, torch.bfloat16, torch.float16)

ALL_CALLBACK_NAMES = ["callback_step", "callback_accept_step", "callback_reject_step"]
ALL_ADJOINT_CALLBACK_NAMES = [name + "_adjoint" for name in ALL_CALLBACK_NAMES]


def _null_callback(*args, **kwargs):
    return None


class Perturb(Enum):
    """Same protocol as the reference's `Perturb` (misc.py:168-171)."""
    NONE = 0
    PREV = 1
    NEXT = 2


def handle_unused_kwargs(solver, unused_kwargs) -> None:
    if len(unused_kwargs) > 0:
        warnings.warn("{}: Unexpected arguments {}".format(solver.__class__.__name__, unused_kwargs))


# ---------------------------------------------------------------------------------------------------
# Built-in norms.  The solvers recognise these objects and evaluate them inside the fused error-norm
# kernel; any other callable is treated as a user norm and receives a materialised tensor.
# ---------------------------------------------------------------------------------------------------
class BuiltinNorm:
    """Marker base: max over the selected segments of sqrt(mean(x^2)) (misc.py:22-33)."""

    def __init__(self, n_skip_tail: int = 0, name: str = "rms"):
        self.n_skip_tail = n_skip_tail   # trailing segments left out of the max ('seminorm')
        self.name = name

    def __call__(self, x):
        # Tensor-level definition, used only if someone calls the object directly (e.g. a user
        # wrapping the default norm).  Runs on whatever device x lives on.
        if isinstance(x, torch.Tensor):
            return x.abs().pow(2).mean().sqrt()
        vals = [xi.abs().pow(2).mean().sqrt() for xi in x if xi.numel() > 0]
        if self.n_skip_tail:
            vals = vals[:len(vals) - self.n_skip_tail]
        return max(vals) if vals else 0.0

    def __repr__(self):
        return f"<torchdiffeq_amd builtin norm '{self.name}'>"


rms_norm = BuiltinNorm(name="rms")        # default for tensor states (misc.py:265)
mixed_norm = BuiltinNorm(name="mixed")    # default for tuple states (misc.py:245)


# ---------------------------------------------------------------------------------------------------
# State layout
# ---------------------------------------------------------------------------------------------------
class FuncOutputTypeError(TypeError, AttributeError):
    """func returned something that is not a Tensor (the reference: AttributeError from `f.shape`, rk_common.py:69)."""


class UnsupportedStateDtype(TypeError, NotImplementedError):
    """An integer / bool state.  The reference raises NotImplementedError from `nextafter` (misc.py:185-196) for the adaptive
    methods and for `perturb=True`; its fixed-grid methods WITHOUT perturbation accept such a state and let type promotion
    turn `y0 + dt * f` into floats.  Deviation: this package refuses integer / bool states for every
    method — there is no kernel for them and no ODE whose state is an integer; convert with `y0.float()`."""


class StateLayout:
    """Flat layout of a (tuple) state: segment s occupies [offset[s], offset[s]+numel[s]) and every
    offset is a multiple of `chunk` elements, so all segments are 16-byte aligned for the vector path
    and no reduction chunk straddles two segments.  A single tensor is one unpadded segment."""

    def __init__(self, shapes: Sequence[torch.Size], is_tuple: bool, chunk: Optional[int] = None):
        self.shapes = [torch.Size(s) for s in shapes]
        self.is_tuple = is_tuple
        self.numels = [int(s.numel()) for s in self.shapes]
        self.chunk = chunk or _native.pick_chunk(max(self.numels) if self.numels else 1)
        self.offsets: List[int] = []
        off = 0
        for n in self.numels:
            self.offsets.append(off)
            off += max(1, math.ceil(n / self.chunk)) * self.chunk if len(self.shapes) > 1 else n
        self.total = max(off, 1) if len(self.shapes) > 1 else (self.numels[0] if self.numels else 0)
        self._chunk_starts = None                      # built on first use by pack_fused
        self._unit_scales = (1.0,) * len(self.shapes)

    @property
    def n_seg(self) -> int:
        return len(self.shapes)

    def pack(self, tensors: Sequence[torch.Tensor], dtype=None, negate: Sequence[bool] = ()) -> torch.Tensor:
        """Copy the components into a fresh flat buffer (a view if the state is a single tensor)."""
        if len(self.shapes) == 1 and not negate:
            t = tensors[0]
            if dtype is not None and t.dtype != dtype:
                t = t.to(dtype)
            return t.reshape(-1).contiguous()
        first = tensors[0]
        # padding zero-filled: every kernel streams over the padding as well, and the time-gradient dot products
        # (tdeq_multi_dot: sum g x over the WHOLE flat vector) would turn a stray NaN there into a NaN gradient
        flat = torch.zeros(self.total, dtype=dtype or first.dtype, device=first.device)
        for i, (t, off, n) in enumerate(zip(tensors, self.offsets, self.numels)):
            dst = flat[off:off + n]
            src = t.reshape(-1)
            if negate and negate[i]:
                torch.neg(src, out=dst) if src.dtype == dst.dtype else dst.copy_(-src)
            else:
                dst.copy_(src)
        return flat

    def pack_fused(self, kernels, pieces: Sequence[Optional[torch.Tensor]], dtype, device,
                   scales: Optional[Sequence[float]] = None) -> torch.Tensor:
        """`pack` as ONE kernel launch per <= TDEQ_INLINE_SEGMENTS consecutive segments (tdeq_pack_segments): piece s
        (None = zeros) times scales[s] (+-1) into segment s of a fresh flat buffer, padding zero-filled.  A model with
        40 parameter tensors packs its augmented state in 3 launches instead of ~90 copy / neg / zero ops.  Falls back
        to `pack` for a single unpadded segment."""
        n = self.n_seg
        if n == 1 or os.environ.get("TDEQ_PACK_FUSED", "1") == "0":
            neg = [sc < 0 for sc in scales] if scales is not None else ()
            filled = [torch.zeros(m, dtype=dtype, device=device) if t is None else t
                      for t, m in zip(pieces, self.numels)]
            return self.pack(filled, dtype=dtype, negate=neg if any(neg) else ())
        # (per-evaluation path of the adjoint's backward solve: the kernel needs only addresses, so no detach / reshape
        # views are made — each costs 1–2 us of host time per piece — and the constant tables are built once)
        srcs = []
        for t, m in zip(pieces, self.numels):
            if t is None or m == 0:
                srcs.append(None)
                continue
            if t.dtype != dtype:
                t = t.detach().to(dtype)
            if t.numel() != m:
                raise RuntimeError(f"func returned a component with {t.numel()} elements where the state has {m}")
            srcs.append(t if t.is_contiguous() else t.contiguous())
        out = torch.empty(self.total, dtype=dtype, device=device)
        G = _native.TDEQ_INLINE_SEGMENTS
        if n <= G:
            if self._chunk_starts is None:
                self._chunk_starts = [off // self.chunk for off in self.offsets]
            kernels.pack_segments(out, srcs, self._chunk_starts, self.numels,
                                  self._unit_scales if scales is None else scales, self.chunk)
            return out
        scales = self._unit_scales if scales is None else list(scales)
        for a in range(0, n, G):           # consecutive segments span a contiguous, chunk-aligned range of `out`
            b = min(a + G, n)
            lo, hi = self.offsets[a], (self.offsets[b] if b < n else self.total)
            kernels.pack_segments(out[lo:hi], srcs[a:b], [(off - lo) // self.chunk for off in self.offsets[a:b]],
                                  self.numels[a:b], scales[a:b], self.chunk)
        return out

    def unpack(self, flat: torch.Tensor, lead: Tuple[int, ...] = (), lo: int = 0,
               hi: Optional[int] = None) -> Tuple[torch.Tensor, ...]:
        """Views of the components [lo, hi) of `flat[..., total]` shaped (*lead, *shape) (a view costs ~7 us of host
        time: callers on a per-evaluation path ask only for the segments they read)."""
        return tuple(flat[..., off:off + n].view((*lead, *shape))
                     for off, n, shape in zip(self.offsets[lo:hi], self.numels[lo:hi], self.shapes[lo:hi]))

    def segments(self, rtol, atol) -> List[Tuple[int, int, float, float]]:
        """[(offset, numel, rtol, atol)] with scalar or per-component tolerances (misc.py:115-123)."""
        rt = _per_segment(rtol, self.n_seg, "rtol")
        at = _per_segment(atol, self.n_seg, "atol")
        return [(off, n, r, a) for off, n, r, a in zip(self.offsets, self.numels, rt, at)]


def pack_differentiable(layout: "StateLayout", tensors: Sequence[torch.Tensor], dtype=None) -> torch.Tensor:
    """Flat (chunk-padded) state built with autograd-visible ops, so gradients flow back to the components."""
    if layout.n_seg == 1:
        t = tensors[0]
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t.reshape(-1).contiguous()
    pieces = []
    for i, t in enumerate(tensors):
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        pieces.append(t.reshape(-1))
        end = layout.offsets[i + 1] if i + 1 < layout.n_seg else layout.total
        pad = end - (layout.offsets[i] + layout.numels[i])
        if pad:
            pieces.append(torch.zeros(pad, dtype=t.dtype, device=t.device))
    return torch.cat(pieces)


def _per_segment(tol, n_seg: int, name: str) -> List[float]:
    if isinstance(tol, torch.Tensor):
        if tol.dim() == 0:
            return [float(tol)] * n_seg
        tol = tol.tolist()
    try:
        vals = [float(torch.as_tensor(v)) for v in tol]
    except TypeError:
        return [float(tol)] * n_seg
    assert len(vals) == n_seg, \
        "If using tupled {} it must have the same length as the tuple y0".format(name)
    return vals


def component_norm(layout: "StateLayout", n_skip_tail: int = 0):
    """The built-in norm (max over the components of sqrt(mean(x^2)), misc.py:22-33) as a plain callable on the flat,
    per-component padded state — for code that cannot go through the segmented norm kernel."""
    def _norm(flat, _lay=layout, _skip=n_skip_tail):
        parts = [c for c in _lay.unpack(flat) if c.numel() > 0]
        if _skip:
            parts = parts[:len(parts) - _skip]
        return max(c.abs().pow(2).mean().sqrt() for c in parts)
    return _norm


def vector_tolerances(rtol, atol, layout: "StateLayout", device, dtype=torch.float64, tuple_entries_too=False):
    """`(rtol, atol)` in the forms the reference's arithmetic sees when a tolerance is given PER ELEMENT — a tensor / list
    that broadcasts against a tensor state, or tuple entries that are vectors over their component — else None.
    The reference needs no code for this: `atol + rtol * max(|y0|, |y1|)` simply broadcasts (misc.py:80-82), the
    tolerances having become tensors of the time dtype W in rk_common.py:186-187 (tuple tolerances: one flat vector,
    misc.py:115-123).  What type promotion then does depends on WHICH of the two is dimensioned — a 0-dim W tensor
    times the fp32 state is an fp32 product, a W vector times it a W product — so each tolerance comes back as it is
    there: a 0-dim tensor, or a vector over the flat (padded) state; the padding gets (0, 1): quotients stay finite.
    `tuple_entries_too`: also for a tuple of SCALAR entries (the literal host path: the reference's tolerance is a flat
    W vector then, its error ratio a W number; the HIP kernels take such entries per segment, in the state's type)."""
    def scalar(tol):
        return torch.as_tensor(tol, dtype=dtype, device=device).reshape(())      # rk_common.py:186-187

    if not layout.is_tuple:
        def form(tol):
            if isinstance(tol, torch.Tensor):
                return tol.dim() > 0
            return isinstance(tol, (list, tuple))
        if not (form(rtol) or form(atol)):
            return None
        shape = layout.shapes[0]
        return tuple(torch.broadcast_to(torch.as_tensor(tol, dtype=dtype, device=device), shape).reshape(-1).contiguous()
                     if form(tol) else scalar(tol) for tol in (rtol, atol))
    # tuple state: a sequence with one entry per component is the ordinary case (scalars: handled per segment by the
    # kernels); only vector ENTRIES need the per-element form
    def entries(tol):
        # `iter(tol)` then `tuple(tol)` (misc.py:113-118): ANY iterable with one entry per component — list, tuple,
        # 1-D tensor, numpy array —; a 0-dim tensor / number is not iterable and stays a scalar
        if isinstance(tol, (str, bytes)) or (isinstance(tol, torch.Tensor) and tol.dim() == 0):
            return None
        try:
            ent = list(tol)
        except TypeError:
            return None
        return ent if len(ent) == layout.n_seg else None

    def is_vector(v):
        # an entry goes through `torch.as_tensor(v)` there (misc.py:121): a tensor, a Python list / tuple of numbers, a
        # numpy array are all vectors over their component when they hold more than one number
        if isinstance(v, torch.Tensor):
            return v.numel() > 1
        if isinstance(v, (int, float)):
            return False
        try:
            return torch.as_tensor(v).numel() > 1
        except Exception:
            return False
    er, ea = entries(rtol), entries(atol)
    has_vec = any(e is not None and any(is_vector(v) for v in e) for e in (er, ea))
    if not (has_vec or (tuple_entries_too and (er is not None or ea is not None))):
        return None
    out = []
    for tol, ent, pad in ((rtol, er, 0.0), (atol, ea, 1.0)):
        if ent is None:
            out.append(scalar(tol))
            continue
        # `torch.as_tensor(tol_).expand(shape.numel())` (misc.py:115-123): a 0-dim / [1] / [n] entry — a Python float is
        # an fp32 number at that point (the default dtype) —; anything else, e.g. an entry shaped like its component,
        # raises there, and therefore here; the pieces are concatenated (promoted) and cast to W
        pieces = [torch.as_tensor(e, device=device).expand(n) for e, n in zip(ent, layout.numels)]
        common = functools.reduce(torch.promote_types, [p_.dtype for p_ in pieces])
        v = torch.full((layout.total,), pad, dtype=dtype, device=device)
        for p_, off, n in zip(pieces, layout.offsets, layout.numels):
            v[off:off + n] = p_.to(common).to(dtype)
        out.append(v)
    return tuple(out)


def empty_solution(ci: "CheckedInputs", like: torch.Tensor):
    """The solution of a state without a single element: `[len(t), *shape]` per component, nothing to integrate (the
    kernels are never asked to run on a null buffer; the reference's fixed-grid solvers return the same, its adaptive
    ones trip over the NaN norm of nothing — docs/LAB_NOTEBOOK.md §8)."""
    rows = [torch.empty((len(ci.t), *shape), dtype=like.dtype, device=like.device) for shape in ci.layout.shapes]
    return tuple(rows) if ci.layout.is_tuple else rows[0]


def plugin_solver_inputs(solver_cls, layout: "StateLayout", options: dict, rtol, atol, device):
    """`(options, rtol, atol)` for constructing `solver_cls`.  This package's own classes take them as they are.  Anything
    else registered in `SOLVERS` that follows the reference's protocol `cls(func=, y0=, rtol=, atol=, **options)
    .integrate(t)` (odeint.py:92, solvers.py:28) — the reference's own classes, a third party's — does its arithmetic with
    torch ops on the flat state and calls `func(t, y, perturb=...)` / `norm(y)` / `func.callback_*` itself (OdeFunc
    answers those calls).  What such a class cannot know is the flat layout of a TUPLE state here — components padded to
    chunk boundaries, tolerances per component — so the built-in norm becomes a callable over the components and
    per-component tolerances are expanded per element, as the reference's `_check_inputs` does (misc.py:237-254)."""
    if getattr(solver_cls, "flat_state_native", False) or not layout.is_tuple:
        return options, rtol, atol
    options = dict(options)
    norm = options.get("norm")
    if isinstance(norm, BuiltinNorm):
        options["norm"] = component_norm(layout, norm.n_skip_tail)

    def per_element(tol, name):
        vals = _per_segment(tol, layout.n_seg, name)
        if len(set(vals)) == 1:
            return vals[0] if isinstance(tol, (list, tuple)) else tol
        out = torch.full((layout.total,), vals[-1], dtype=torch.float64, device=device)
        for off, n, v in zip(layout.offsets, layout.numels, vals):
            out[off:off + n] = v
        return out
    return options, per_element(rtol, "rtol"), per_element(atol, "atol")


# ---------------------------------------------------------------------------------------------------
# func wrapper
# ---------------------------------------------------------------------------------------------------
class OdeFunc:
    """The user's `func(t, y)` as the solvers see it: flat state in, flat contiguous state out.

    `eval(t, y_flat, perturb)` takes the (ascending, possibly negated) solver time as a host scalar,
    applies the reference's `_PerturbFunc` / `_ReverseFunc` time semantics (misc.py:158-197) on the
    host, and hands the user a 0-dim device tensor.  Outputs are NOT multiplied by the time sign —
    the kernels fold it into `dt`.  Calling the object like the reference's wrapped func,
    `f(t_tensor, y_flat, perturb=...)`, is also supported (third-party solver classes).
    """

    def __init__(self, base_func: Callable, layout: StateLayout, sign: float, dtype: torch.dtype,
                 device: torch.device):
        self.base_func = base_func
        self.layout = layout
        self.sign = float(sign)
        self.dtype = dtype
        self.device = device
        # T = y0.abs().dtype: the precision of every time-like scalar func sees (misc.py:185; real for complex states)
        self.np_dtype = scalar_type(dtype)          # host stand-in for 0-dim tensors of that type (_scalars.py)
        self.time_dtype = real_dtype(dtype)
        self.nfe = 0
        self.grad_output_seen = False   # an evaluation in grad mode returned a tensor that is part of an autograd graph
        self._kernels = None
        self._anchor_user = None     # user-time t[0] when `t` requires grad (adaptive solvers)
        self.strict_numel = False    # see _conform; set by check_inputs from the solver class
        for name in ALL_CALLBACK_NAMES:
            setattr(self, name, _null_callback)

    def kernels(self):
        """The HIP kernel interface of the state's device (created on first use)."""
        if self._kernels is None:
            self._kernels = _native.get_kernels(self.device, self.dtype)
        return self._kernels

    def graph_key(self):
        """Extra identity of this wrapper for the captured-step cache (subclasses that compute more than
        `base_func` — the adjoint's augmented dynamics — say what else a captured step depends on)."""
        return ()

    def set_time_anchor(self, anchor) -> None:
        """`anchor` = t[0] in solver time (a 0-dim tensor in the autograd graph of `t`) or None.  Every time the
        adaptive solvers hand to func is t[0] + constants, so its gradient flows to this anchor
        (rk_common.py:72-78 with t0 = t[0] + sum of detached step sizes)."""
        self._anchor_user = None if anchor is None else anchor * self.sign

    # -- time handling -----------------------------------------------------------------------------
    def user_time(self, t, perturb: Perturb = Perturb.NONE) -> float:
        """Solver time -> the value the user's func sees (state-precision, perturbed, un-negated)."""
        tt = self.np_dtype(t)
        if perturb is Perturb.NEXT:
            tt = nextafter(tt, tt + 1)
        elif perturb is Perturb.PREV:
            tt = nextafter(tt, tt - 1)
        return float(self.sign * tt)

    def time_tensor(self, value: float, shadow=None) -> torch.Tensor:
        """0-dim tensor with the host value; its gradient goes to `shadow` (user time) or the anchor."""
        v = torch.full((), value, dtype=self.time_dtype, device=self.device)
        return stitch(v, shadow if shadow is not None else self._anchor_user)

    def time_tensors(self, kernels, times_and_perturbs, shadows=None) -> Tuple[torch.Tensor, ...]:
        """0-dim device tensors for several evaluation times with ONE launch (instead of one fill kernel
        per stage): the values are computed on the host and travel in the kernel arguments."""
        vals = [self.user_time(t, p) for t, p in times_and_perturbs]
        buf = torch.empty(len(vals), dtype=self.time_dtype, device=self.device)
        kernels.fill_scalars(buf, vals)
        out = buf.unbind(0)
        if shadows is None and self._anchor_user is None:
            return out
        if shadows is None:
            shadows = [self._anchor_user] * len(out)
        return tuple(stitch(v, sh) for v, sh in zip(out, shadows))

    def eval_at(self, t_user: torch.Tensor, y_flat: torch.Tensor) -> torch.Tensor:
        """Evaluate with a time tensor produced by `time_tensors`."""
        self.nfe += 1
        return self.call_base(t_user, y_flat)

    # -- evaluation --------------------------------------------------------------------------------
    def eval(self, t, y_flat: torch.Tensor, perturb: Perturb = Perturb.NONE, shadow=None) -> torch.Tensor:
        assert isinstance(perturb, Perturb), "perturb argument must be of type Perturb enum"
        self.nfe += 1
        t_user = self.time_tensor(self.user_time(t, perturb), shadow)
        return self.call_base(t_user, y_flat)

    def _conform(self, f, shape: torch.Size, what: str) -> torch.Tensor:
        """func's output as the kernels need it: on the state's device, with the state's element count.  Accepted is
        what the reference's own arithmetic accepts, no more — a user bug must not be integrated silently:
        * tensor state, fixed-grid methods: anything that BROADCASTS to the state shape (0-dim, [1], a row, extra
          leading 1-dims), expanded as `y0 + dt * f` expands it (rk_common.py:110-157, fixed_grid.py);
        * tensor state, adaptive methods (`strict_numel`): the stage buffer takes func's shape and is viewed as the
          state (rk_common.py:69-79, 366), so the output must broadcast AND have the state's element count;
        * tuple state: every component is flattened into the state vector (misc.py:145), so its element count must be
          the component's (any shape).
        Everything else raises RuntimeError here (the reference: a broadcasting / view error from inside the step)
        instead of being read out of bounds by a kernel."""
        if not isinstance(f, torch.Tensor):
            if isinstance(f, (int, float)) and not isinstance(f, bool) and not self.strict_numel and not what:
                f = torch.tensor(f, dtype=self.dtype, device=self.device)   # `y0 + dt * 1.0` is fine in fixed_grid.py
            else:
                raise FuncOutputTypeError("func must return a Tensor{}; got {}".format(what, type(f).__name__))
        if f.device != self.device and f.dim() == 0:
            f = f.to(self.device)       # `y0 + dt * f` accepts a 0-dim tensor from another device (rk_common.py:79)
        if f.device != self.device:
            raise RuntimeError("func returned a tensor on '{}'{} but the state lives on '{}'".format(
                f.device, what, self.device))
        if f.shape == shape:
            return f
        if self.layout.is_tuple:
            if f.numel() != shape.numel():
                raise RuntimeError("func returned shape {}{} ({} elements) for a state component of shape {} ({} elements)"
                                   .format(tuple(f.shape), what, f.numel(), tuple(shape), shape.numel()))
            return f
        lead = f.dim() - len(shape)
        try:
            ok = torch.broadcast_shapes(f.shape, shape) == ((1,) * lead + tuple(shape) if lead > 0 else tuple(shape))
        except RuntimeError:
            ok = False
        if ok and self.strict_numel and f.numel() != shape.numel():
            ok = False
        if not ok:
            raise RuntimeError("func returned shape {}{} which {} the state shape {}".format(
                tuple(f.shape), what, "does not match" if self.strict_numel else "does not broadcast to", tuple(shape)))
        return f.reshape(f.shape[lead:]).expand(shape) if lead > 0 else f.expand(shape)

    def call_base(self, t_user: torch.Tensor, y_flat: torch.Tensor) -> torch.Tensor:
        lay = self.layout
        grad = torch.is_grad_enabled()
        if lay.is_tuple:
            f = self.base_func(t_user, lay.unpack(y_flat))
            if len(f) != lay.n_seg:
                raise RuntimeError("func returned {} components for a state of {}".format(len(f), lay.n_seg))
            f = tuple(self._conform(f_, shape, " (component {})".format(i))
                      for i, (f_, shape) in enumerate(zip(f, lay.shapes)))
            if grad and any(f_.requires_grad for f_ in f):
                out = pack_differentiable(lay, f, self.dtype)     # backprop through the solver
            else:
                out = lay.pack_fused(self.kernels(), f, self.dtype, self.device)
        else:
            f = self._conform(self.base_func(t_user, y_flat.view(lay.shapes[0])), lay.shapes[0], "")
            if f.dtype != self.dtype:
                f = f.to(self.dtype)
            out = f.reshape(-1)
            if not out.is_contiguous():
                out = out.contiguous()
        if out.requires_grad:
            if not grad:
                out = out.detach()      # func built its own graph internally (e.g. a Jacobian trace): drop it
            else:
                # the solve is part of an autograd graph through func's own parameters — seen by the captured-step paths,
                # which write raw buffers (fixed.py `_integrate_graph`: the dynamic guard behind the static look at func)
                self.grad_output_seen = True
        return out

    def __call__(self, t, y_flat, *, perturb: Perturb = Perturb.NONE):
        # Reference-style call: `t` is a 0-dim tensor in (negated) solver time.
        if not isinstance(perturb, Perturb):
            # a solver class written against ANOTHER copy of the enum (the reference's own classes work as plug-ins in
            # SOLVERS): members are matched by name
            perturb = Perturb.__members__.get(getattr(perturb, "name", None), perturb)
        return self.eval(float(t), y_flat, perturb) * self.sign if self.sign != 1.0 else \
            self.eval(float(t), y_flat, perturb)


# ---------------------------------------------------------------------------------------------------
# Input checks
# ---------------------------------------------------------------------------------------------------
def _assert_floating(name, t):
    if not torch.is_floating_point(t):
        raise TypeError("`{}` must be a floating point Tensor but is a {}".format(name, t.type()))


def check_timelike(name, timelike, can_grad):
    assert isinstance(timelike, torch.Tensor), "{} must be a torch.Tensor".format(name)
    _assert_floating(name, timelike)
    assert timelike.ndimension() == 1, "{} must be one dimensional".format(name)
    if not can_grad:
        assert not timelike.requires_grad, "{} cannot require gradient".format(name)
    diff = timelike[1:] > timelike[:-1]
    assert diff.all() or (~diff).all(), "{} must be strictly increasing or decreasing".format(name)


def _flip_option(options, name):
    value = options.get(name)
    if isinstance(value, torch.Tensor):
        options[name] = -value


def combine_event_functions(event_fn, t0, y0):
    """event_handling.py:23-35: make every component of a multivariate event function initially positive
    and combine them with a min, so one sign change of the combined scalar marks the first event."""
    with torch.no_grad():
        orientation = event_fn(t0, y0).sign()
    return lambda t, y: torch.min(event_fn(t, y) * orientation)


def find_event(interp_fn, sign0, t0, t1, event_fn, tol: float, time_tensor, scalar=np.float64):
    """Bisection on the step's interpolant (event_handling.py:5-20).  Times are host scalars of type `scalar`
    (fp64 for the adaptive solvers, the state dtype for the fixed-grid ones — solvers.py:132) and every
    operation is rounded in that type, like the reference's 0-dim tensor arithmetic.
    `interp_fn(t) -> y_flat` evaluates the dense output (a kernel launch); `event_fn(t_tensor, y_flat)` is the
    user's scalar event function, whose sign is read back once per iteration (inherent to bisection)."""
    t0, t1 = scalar(t0), scalar(t1)
    with torch.no_grad():
        with np.errstate(all="ignore"):
            width = scalar(scalar(t1 - t0) / scalar(tol))
            # (a 16-bit time type has no numpy logarithm: ATen takes it in fp32 and rounds once)
            log_w = np.log(width) if isinstance(width, np.floating) else scalar(np.log(np.float32(float(width))))
            nitrs = np.ceil(np.float64(log_w / scalar(math.log(2.0))))
        if np.isinf(nitrs) and nitrs > 0:
            raise OverflowError("find_event: cannot bisect to a tolerance of 0 (atol must be positive)")
        nitrs = 0 if np.isnan(nitrs) else int(nitrs)
        for _ in range(nitrs):
            t_mid = scalar(scalar(t1 + t0) / scalar(2.0))
            y_mid = interp_fn(t_mid)
            sign_mid = float(torch.sign(event_fn(time_tensor(t_mid), y_mid)).detach())
            if sign0 == sign_mid:
                t0 = t_mid
            else:
                t1 = t_mid
        event_t = scalar(scalar(t0 + t1) / scalar(2.0))
    return event_t, interp_fn(event_t)


class CheckedInputs:
    """Result of `check_inputs`: everything `odeint` needs to build and run a solver."""
    __slots__ = ("layout", "func", "y0_flat", "t", "rtol", "atol", "method", "options", "event_fn",
                 "t_is_reversed", "original_func")


def check_inputs(func, y0, t, rtol, atol, method, options, event_fn, SOLVERS) -> CheckedInputs:
    """Normalise `(func, y0, t, ...)`; same accept/reject behaviour as misc.py:200-345."""
    if event_fn is not None:
        if len(t) != 2:
            raise ValueError(f"We require len(t) == 2 when in event handling mode, but got len(t)={len(t)}.")
        # multivariate event functions: all components made initially positive, combined by a min
        event_fn = combine_event_functions(event_fn, t[0], y0)

    original_func = func
    is_tuple = not isinstance(y0, torch.Tensor)
    if is_tuple:
        assert isinstance(y0, tuple), "y0 must be either a torch.Tensor or a tuple"
        for y0_ in y0:
            assert isinstance(y0_, torch.Tensor), "y0 must be either a torch.Tensor or a tuple"
        shapes = [y0_.shape for y0_ in y0]
        first = y0[0]
        # components of different dtypes: the reference concatenates them (misc.py:206-207), i.e. the whole state —
        # what func is given and what is returned — has the promoted dtype
        dtype = functools.reduce(torch.promote_types, [y0_.dtype for y0_ in y0]) if len(y0) else None
    else:
        shapes = [y0.shape]
        first = y0
        dtype = first.dtype
    device = first.device
    # float32 / float64 (and complex64 / complex128 through their real views) and — r05 — bfloat16 / float16 states run on
    # the HIP kernels, each integrated in its own precision as the reference does (misc.py:185-187)
    if dtype not in SUPPORTED_STATE_DTYPES:
        raise UnsupportedStateDtype("torchdiffeq_amd supports float32 / float64 / complex64 / complex128 states (and "
                                    f"bfloat16 / float16), got {dtype}; convert an integer state with y0.float()")

    if options is None:
        options = {}
    else:
        options = options.copy()
    if method is None:
        method = "dopri5"
    if method not in SOLVERS:
        raise ValueError('Invalid method "{}". Must be one of {}'.format(
            method, '{"' + '", "'.join(SOLVERS.keys()) + '"}.'))

    layout = StateLayout(shapes, is_tuple)
    if is_tuple:
        y0_flat = layout.pack([y_.detach() for y_ in y0], dtype=dtype)
        if "norm" in options:
            user_norm = options["norm"]
            if not isinstance(user_norm, BuiltinNorm):
                def _norm(tensor, _user=user_norm, _lay=layout):
                    return _user(_lay.unpack(tensor))
                options["norm"] = _norm
        else:
            options["norm"] = mixed_norm
    else:
        y0_flat = y0.detach().reshape(-1).contiguous()
        if "norm" not in options:
            options["norm"] = rms_norm
        elif not isinstance(options["norm"], BuiltinNorm):
            # the user's norm sees the state in ITS shape (the reference never flattens a tensor state)
            def _norm(tensor, _user=options["norm"], _shape=shapes[0]):
                return _user(tensor.view(_shape))
            options["norm"] = _norm

    check_timelike("t", t, True)
    t_is_reversed = bool(len(t) > 1 and t[0] > t[1])
    if t_is_reversed:
        t = -t
        if "grid_constructor" in options:
            _gc = options["grid_constructor"]
            options["grid_constructor"] = lambda func, y0, t: -_gc(func, y0, -t)
        _flip_option(options, "step_t")
        _flip_option(options, "jump_t")
    assert (t[1:] > t[:-1]).all(), "t must be strictly increasing or decreasing"

    if torch.is_tensor(rtol):
        assert not rtol.requires_grad, "rtol cannot require gradient"
    if torch.is_tensor(atol):
        assert not atol.requires_grad, "atol cannot require gradient"

    if t.device != device:
        warnings.warn("t is not on the same device as y0. Coercing to y0.device.")
        t = t.to(device)

    wrapped = OdeFunc(func, layout, -1.0 if t_is_reversed else 1.0, dtype, device)
    wrapped.strict_numel = bool(getattr(SOLVERS[method], "func_output_numel_must_match", False))
    if event_fn is not None:
        # the solvers call event_fn(t, y_flat) with t a 0-dim tensor in (ascending) solver time
        # (misc.py:137-165: _TupleInputOnlyFunc, _ReverseFunc)
        _user_event = event_fn
        if is_tuple and t_is_reversed:
            event_fn = lambda t_, y_, _e=_user_event, _lay=layout: _e(-t_, _lay.unpack(y_))
        elif is_tuple:
            event_fn = lambda t_, y_, _e=_user_event, _lay=layout: _e(t_, _lay.unpack(y_))
        elif t_is_reversed:
            event_fn = lambda t_, y_, _e=_user_event, _shape=shapes[0]: _e(-t_, y_.view(_shape))
        else:
            event_fn = lambda t_, y_, _e=_user_event, _shape=shapes[0]: _e(t_, y_.view(_shape))

    # Callbacks: attributes of the user's func, re-bound to the wrapped func (misc.py:313-343).
    callback_names = set()
    for name in ALL_CALLBACK_NAMES:
        callback = getattr(original_func, name, None)
        if callback is None or callback is _null_callback:
            continue
        callback_names.add(name)
        if is_tuple:
            def callback(t0, y0, dt, _callback=callback, _lay=layout):
                return _callback(t0, _lay.unpack(y0), dt)
        elif len(shapes[0]) != 1:
            # a tensor state reaches the user's callback in ITS shape (the reference never flattens it)
            def callback(t0, y0, dt, _callback=callback, _shape=shapes[0]):
                return _callback(t0, y0.view(_shape), dt)
        if t_is_reversed:
            def callback(t0, y0, dt, _callback=callback):
                return _callback(-t0, y0, dt)
        setattr(wrapped, name, callback)
    for name in ALL_ADJOINT_CALLBACK_NAMES:
        callback = getattr(original_func, name, None)
        if callback is not None:
            setattr(wrapped, name, callback)

    invalid = callback_names - SOLVERS[method].valid_callbacks()
    if len(invalid) > 0:
        warnings.warn("Solver '{}' does not support callbacks {}".format(method, invalid))

    out = CheckedInputs()
    out.layout, out.func, out.y0_flat, out.t = layout, wrapped, y0_flat, t
    out.rtol, out.atol, out.method, out.options = rtol, atol, method, options
    out.event_fn, out.t_is_reversed, out.original_func = event_fn, t_is_reversed, original_func
    return out
