"""Host-side scalar types: the time-like 0-dim tensors of the reference as host numbers that round like them.

The reference keeps t0, t1, dt, tolerances and controller constants as 0-dim tensors on the state's device; this
package keeps them on the host (docs/LAB_NOTEBOOK.md §2).  For that to be invisible every host operation has to round as the
tensor operation it stands for:

* fp64 / fp32 (`y0.abs().dtype` of real and complex states, and the solver option `dtype`,
  torchdiffeq/_impl/rk_common.py:176-194): numpy's scalar types — IEEE arithmetic in the type itself, Python numbers
  mixing in as weak operands, exactly ATen's promotion for 0-dim tensors.
* bf16 / fp16 (states below fp32, which the reference integrates in their own precision: misc.py:185-187,
  rk_common.py:61-65): numpy has no bfloat16 and rounds a Python operand to float16 BEFORE the operation, whereas
  ATen computes reduced-precision operations in float32 ("opmath") — a Python number or a 0-dim partner is taken at
  float32 precision — and rounds the result once.  `BFloat16Scalar` / `Float16Scalar` do that.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def _f32_to_bf16(x: np.float32) -> float:
    """Round-to-nearest-even of a float32 to bfloat16 (c10::BFloat16's conversion), returned as a Python float."""
    if x != x:
        return math.nan
    u = int(np.float32(x).view(np.uint32))
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return float(np.uint32(u & 0xFFFFFFFF).view(np.float32))


def _f32_to_f16(x: np.float32) -> float:
    with np.errstate(over="ignore"):
        return float(np.float16(x))


def _aten_sqrt(x: float, dtype: torch.dtype) -> float:
    """ATen's own CPU sqrt (its vectorised routine is not the correctly rounded libm one: ~1 % of fp64 arguments differ
    in the last bit) — asked directly, for the one place a host scalar needs it (`x ** 0.5` is dispatched to sqrt)."""
    return float(torch.sqrt(torch.tensor(x, dtype=dtype)))


class _LowScalar(float):
    """A Python float whose value is representable in a 16-bit floating type, standing in for a 0-dim tensor of that
    type.  `a op b` = round_low(float32(a) op float32(b)); a numpy scalar partner is a WIDER 0-dim tensor and wins the
    promotion (the operation is numpy's)."""
    __slots__ = ()
    __array_ufunc__ = None      # numpy scalars defer their binary operators to this class (else `np.float32(2) * x` would take
    #                             x for a Python float and hand back a float64 — 0-dim fp32 x 0-dim bf16 is fp32 in ATen)
    _round = staticmethod(_f32_to_bf16)
    torch_dtype = None
    eps_bits = 0            # explicit significand bits (for nextafter)
    min_exp = 0             # exponent of the smallest normal number

    def __new__(cls, x=0.0):
        if type(x) is cls:
            return x
        with np.errstate(over="ignore"):
            return float.__new__(cls, cls._round(np.float32(x)))

    # -- arithmetic ---------------------------------------------------------------------------------------------
    # What ATen's CPU kernels do with a Python number next to a reduced-precision tensor (measured on torch 2.10,
    # tools/lowfloat_semantics.py): add / sub round it to the tensor's type first; mul and `tensor / number` take it
    # at float32; `number / tensor` is `tensor.reciprocal() * number` (Tensor.__rtruediv__); `tensor ** number`
    # rounds the exponent first.
    def _op(self, other, fn, reflected: bool, weak_at_f32: bool):
        if isinstance(other, np.floating):
            a = type(other)(float(self)) if other.dtype.itemsize >= 4 else np.float32(float(self))
            return fn(other, a) if reflected else fn(a, other)
        if not isinstance(other, (float, int)):
            return NotImplemented
        if not weak_at_f32 and not isinstance(other, _LowScalar):
            other = type(self)(other)
        a, b = np.float32(float(self)), np.float32(float(other))
        with np.errstate(all="ignore"):
            r = fn(b, a) if reflected else fn(a, b)
        return type(self)(r)

    def __add__(self, o):
        return self._op(o, lambda a, b: a + b, False, False)

    def __radd__(self, o):
        return self._op(o, lambda a, b: a + b, True, False)

    def __sub__(self, o):
        return self._op(o, lambda a, b: a - b, False, False)

    def __rsub__(self, o):
        return self._op(o, lambda a, b: a - b, True, False)

    def __mul__(self, o):
        return self._op(o, lambda a, b: a * b, False, True)

    def __rmul__(self, o):
        return self._op(o, lambda a, b: a * b, True, True)

    def __truediv__(self, o):
        return self._op(o, lambda a, b: a / b, False, True)

    def __rtruediv__(self, o):
        if isinstance(o, np.floating):
            return self._op(o, lambda a, b: a / b, True, True)
        return (type(self)(1.0) / self) * o

    def __pow__(self, o, mod=None):
        if not isinstance(o, np.floating) and float(o) == 0.5:
            return type(self)(_aten_sqrt(float(self), self.torch_dtype))
        return self._op(o, lambda a, b: a ** b, False, False)

    def __rpow__(self, o, mod=None):
        return self._op(o, lambda a, b: a ** b, True, False)

    def __neg__(self):
        return type(self)(-float(self))

    def __pos__(self):
        return self

    def __abs__(self):
        return type(self)(abs(float(self)))

    def __repr__(self):
        return "{}({})".format(type(self).__name__, float(self))

    @property
    def dtype(self):
        return self.torch_dtype

    # -- neighbours ---------------------------------------------------------------------------------------------
    def spacing(self) -> float:
        """Distance to the next representable number of larger magnitude."""
        v = abs(float(self))
        if v == 0.0 or not math.isfinite(v):
            return 2.0 ** (self.min_exp - self.eps_bits)
        e = max(math.frexp(v)[1] - 1, self.min_exp)
        return 2.0 ** (e - self.eps_bits)

    def nextafter(self, toward) -> "_LowScalar":
        v, target = float(self), float(toward)
        if v != v or target != target:
            return type(self)(math.nan)
        if v == target:
            return self
        if v == 0.0:
            tiny = 2.0 ** (self.min_exp - self.eps_bits)
            return type(self)(tiny if target > 0 else -tiny)
        up = target > v                                  # move towards +inf ?
        away = (up and v > 0) or (not up and v < 0)      # magnitude grows ?
        if away:
            step = self.spacing()
        else:
            # magnitude shrinks: at a power of two the spacing below is half the spacing above
            m, _ = math.frexp(abs(v))
            step = self.spacing() / 2 if (m == 0.5 and abs(v) > 2.0 ** self.min_exp) else self.spacing()
        return type(self)(v + step if up else v - step)


class BFloat16Scalar(_LowScalar):
    __slots__ = ()
    _round = staticmethod(_f32_to_bf16)
    torch_dtype = torch.bfloat16
    eps_bits = 7
    min_exp = -126


class Float16Scalar(_LowScalar):
    __slots__ = ()
    _round = staticmethod(_f32_to_f16)
    torch_dtype = torch.float16
    eps_bits = 10
    min_exp = -14


_SCALAR_TYPES = {torch.float64: np.float64, torch.float32: np.float32, torch.complex128: np.float64,
                 torch.complex64: np.float32, torch.bfloat16: BFloat16Scalar, torch.float16: Float16Scalar}
_REAL_DTYPES = {torch.complex128: torch.float64, torch.complex64: torch.float32}


def real_dtype(dtype: torch.dtype) -> torch.dtype:
    """`y0.abs().dtype` (misc.py:185, rk_common.py:61)."""
    return _REAL_DTYPES.get(dtype, dtype)


def scalar_type(dtype: torch.dtype):
    """The host scalar type standing for 0-dim tensors of `real_dtype(dtype)`."""
    try:
        return _SCALAR_TYPES[dtype]
    except KeyError:
        raise TypeError("torchdiffeq_amd: no host scalar type for {}".format(dtype)) from None


def is_low(scalar_cls) -> bool:
    return isinstance(scalar_cls, type) and issubclass(scalar_cls, _LowScalar)


def nextafter(x, toward):
    """`torch.nextafter` / misc.py:358-361 for a host scalar of any of the types above."""
    if isinstance(x, _LowScalar):
        return x.nextafter(toward)
    return np.nextafter(x, type(x)(toward))


def operand(T, value) -> float:
    """A 0-dim tensor / Python number `value` as the SECOND operand of an elementwise operation on a tensor of scalar
    type T: rounded to T when T is fp32 / fp64 (the common dtype), taken at float32 for the reduced types (opmath)."""
    if is_low(T):
        return float(np.float32(value))
    return float(T(value))


def rdiv(c: float, x):
    """`c / x` for a Python number c and a host scalar x standing for a 0-dim tensor: torch evaluates it as
    `x.reciprocal() * c` (Tensor.__rtruediv__) — two roundings in x's type."""
    if isinstance(x, _LowScalar):
        return c / x
    T = type(x)
    with np.errstate(all="ignore"):
        return T(T(T(1.0) / x) * T(c))


def power(x, e: float):
    """`x ** e` for a Python exponent: ATen's CPU pow(Tensor, Scalar) — ITS sqrt for e = 0.5; float32 bases are raised in
    double and rounded once; reduced types round the exponent to their own precision first."""
    if isinstance(x, _LowScalar):
        return x ** e
    T = type(x)
    with np.errstate(all="ignore"):
        if e == 0.5:
            return T(_aten_sqrt(float(x), torch.float32 if T is np.float32 else torch.float64))
        return T(np.float64(x) ** np.float64(e))
