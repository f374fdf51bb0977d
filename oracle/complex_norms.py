"""CPU restatement of the reference's tolerance-scaled norms for COMPLEX states — the checker of the complex norm
kernels (torchdiffeq_amd/csrc/tdeq_kernels_complex.hpp).  Test infrastructure: never imported by the product.

The reference needs no code of its own for complex states: `y0.abs()` is the complex modulus and `err / tol` a complex
tensor divided by a real one (torchdiffeq/_impl/misc.py:80-82, 50-56, 68; rk_common.py:61 casts the time-like scalars
to `y0.abs().dtype`).  Restated here with the same ATen operations on whatever device the inputs live on — on the CPU
it is the oracle, on the GPU the bit-exact twin of the kernels (ATen's |z| is the device libm's hypot there, which a
CPU cannot reproduce bit for bit: tools/complex_abs_probe.py)."""
from typing import List, Optional, Sequence, Tuple

import torch


def _real_dtype(dtype):
    return torch.float32 if dtype == torch.complex64 else torch.float64


def error_estimate(ks: Sequence[torch.Tensor], coefs: Sequence[float], dt: float, partial: Optional[torch.Tensor] = None):
    """err = (c_0 k_0 + c_1 k_1) + ... [continuing `partial`], c_j = fl_T(fl_T(coef_j) * fl_T(dt)), products and sums
    rounded separately per component, left to right (rk_common.py:89, 201-205)."""
    T = _real_dtype(ks[0].dtype if ks else partial.dtype)
    dtT = torch.tensor(dt, dtype=T)
    acc = partial
    for k, c in zip(ks, coefs):
        cT = float(torch.tensor(c, dtype=T) * dtT)
        term = torch.view_as_complex(torch.view_as_real(k) * cT)      # complex * real scalar: per component
        acc = term if acc is None else torch.view_as_complex(torch.view_as_real(acc) + torch.view_as_real(term))
    return acc


def error_ratio_parts(err, y0, y1, rtol: float, atol: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(r = err / tol, tol) with tol = atol + rtol * max(|y0|, |y1|) — misc.py:80-81 — in T = the real type."""
    T = _real_dtype(y0.dtype)
    tol = float(torch.tensor(atol, dtype=T)) + float(torch.tensor(rtol, dtype=T)) * torch.max(y0.abs(), y1.abs())
    return err / tol, tol


def segment_sums(r: torch.Tensor, segs: Sequence[Tuple[int, int]]) -> List[float]:
    """sum |r|^2 per segment (element offset, numel) in fp64, |r|^2 = re^2 + im^2 formed in double (misc.py:22 squares the
    rounded modulus and accumulates in T; the kernels' fp64 accumulation takes the components — docs/LAB_NOTEBOOK.md §8)."""
    out = []
    for off, n in segs:
        v = torch.view_as_real(r[off:off + n]).double()
        out.append(float((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]).sum()))
    return out


def init_quotients(mode: int, a, b, y, rtol: float, atol: float):
    """misc.py:50-56,68: scale = atol + |y| * rtol; mode 0: (a / scale, b / scale); mode 1: ((a - b) / scale, None)."""
    T = _real_dtype(y.dtype)
    scale = float(torch.tensor(atol, dtype=T)) + y.abs() * float(torch.tensor(rtol, dtype=T))
    if mode == 0:
        return a / scale, b / scale
    return (a - b) / scale, None
