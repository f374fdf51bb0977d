"""Hunt for the accept / reject flip: the kernels accumulate the error norm in fp64 and form the
embedded error left to right, the reference accumulates in the state dtype in ATen's blocked order
(torchdiffeq/_impl/misc.py:22-23,80-82; rk_common.py:89).  A trial step whose error ratio lies within rounding of 1.0
can therefore be accepted by one and rejected by the other.  This test manufactures such steps — random dopri5 stage
data, tolerances rescaled so that the exact ratio is 1 + delta for delta from 3e-6 down to 0 on both sides — and bounds
the effect:

  * the kernel's ratio (fp32-rounded, as the solver uses it) is within 1e-6 relative of the exact (fp64) ratio and of the
    reference-style fp32 ratio, everywhere;
  * outside the band |ratio - 1| <= 1e-6 the two decisions ALWAYS agree; inside it they may differ (the reference
    differs from itself there when only its thread count changes) — the number of such flips is reported.
"""
import numpy as np
import pytest
import torch

from torchdiffeq_amd import _native
from torchdiffeq_amd.tableaus import DOPRI5, SparseRow

DELTAS = [s * d for d in (3e-6, 2e-6, 1e-6, 3e-7, 1e-7, 3e-8, 1e-8) for s in (1.0, -1.0)] + [0.0]


def _reference_style_ratio(y0, y1, ks, c_err, dt, rtol, atol):
    """The reference's own op sequence in fp32 (stage-minor k, sum over the stage axis, fp32 RMS)."""
    k = torch.stack(ks, dim=-1)                                   # [N, 7], as rk_common.py:69
    coef = (torch.tensor(c_err, dtype=torch.float64).to(y0.dtype) * torch.tensor(dt, dtype=torch.float64).to(y0.dtype)).to(y0.device)
    err = torch.sum(k * coef, dim=-1)                             # rk_common.py:89
    tol = atol + rtol * torch.max(y0.abs(), y1.abs())             # misc.py:81 (0-dim fp64 scalars: stays fp32)
    q = err / tol
    exact = float(q.double().pow(2).mean().sqrt())                # the same quotients, accumulated exactly enough
    return float(q.abs().pow(2).mean().sqrt()), exact


def _hunt(kern, device, n, seeds):
    c_err = list(DOPRI5.c_error)
    row = SparseRow.from_dense(DOPRI5.c_error)
    flips, worst_rel, checked = [], 0.0, 0
    for seed in seeds:
        g = torch.Generator().manual_seed(seed)
        y0 = torch.randn(n, generator=g).to(device)
        y1 = (y0.cpu() + 0.01 * torch.randn(n, generator=g)).to(device)
        ks = [torch.randn(n, generator=g).to(device) for _ in range(7)]
        dt, rtol, atol = 0.0371, 1e-3, 1e-4
        _, r0 = _reference_style_ratio(y0, y1, ks, c_err, dt, rtol, atol)
        for delta in DELTAS:
            s = r0 * (1.0 + delta)          # tolerances scaled by s: the ratio becomes ~ 1 / (1 + delta)
            rt, at = rtol * s, atol * s
            plan = kern.make_plan([(0, n, rt, at)], n, _native.pick_chunk(n), device)
            kern.error_norm(plan, y0, y1, [ks[j] for j in row.idx], row.coef, dt)
            sumsq, _, bad = kern.read_norms(plan)
            assert bad == [0.0]
            ratio_k = float(np.float32(np.sqrt(sumsq[0] / n)))    # solvers._segment_norm
            ratio_ref, exact = _reference_style_ratio(y0, y1, ks, c_err, dt, rt, at)
            worst_rel = max(worst_rel, abs(ratio_k - exact) / exact, abs(ratio_k - ratio_ref) / ratio_ref)
            checked += 1
            if (ratio_k <= 1.0) != (ratio_ref <= 1.0):
                flips.append((seed, delta, ratio_k, ratio_ref, exact))
                assert abs(exact - 1.0) <= 1e-6, ("decisions differ OUTSIDE the rounding band", flips[-1])
    return flips, worst_rel, checked


@pytest.mark.gpu
def test_accept_reject_flip_band_gpu(hip_kernels):
    flips, worst_rel, checked = _hunt(hip_kernels, torch.device("cuda:0"), 1 << 20, range(6))
    print(f"{len(flips)} flips in {checked} near-threshold steps; worst relative ratio difference {worst_rel:.2e}")
    assert worst_rel < 1e-6
    assert len(flips) <= checked // 4


def test_accept_reject_flip_band_oracle(oracle_kernels):
    """The same hunt with the CPU oracle (identical arithmetic to the kernels) — runs in the build container."""
    flips, worst_rel, checked = _hunt(oracle_kernels, torch.device("cpu"), 1 << 16, range(4))
    print(f"{len(flips)} flips in {checked} near-threshold steps; worst relative ratio difference {worst_rel:.2e}")
    assert worst_rel < 1e-6
    assert len(flips) <= checked // 4
