"""oracle/lp_kernels.py (the CPU restatement for bfloat16 / float16 states) against vectors the REFERENCE produced on
reduced-precision tensors (tests/golden/make_golden_lowp.py -> lowp_kernels.npz), and the package's torch-op host path
(`_fallback.LowPrecisionHostKernels`, bit-identical to the reference on the CPU) against the same oracle with the rows
summed in the kernels' order — so that "HIP == oracle bit for bit" on the GPU (tests/test_lowp_gpu.py) means "HIP == the
reference's arithmetic up to the order of a row's float32 accumulation".

Where the oracle and the reference can differ: a tableau row summed over its non-zero weights left to right vs ATen's
own order over the dense row — float32 accumulation of <= 14 products that carry 8 (bf16) / 11 (fp16) significand bits,
i.e. exact in almost every element; the bound below is one unit in the last place of the storage type on a small
fraction of the elements, and identical everywhere else."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import lp_kernels as olp  # noqa: E402
from torchdiffeq_amd import _fallback, tableaus as tb  # noqa: E402

Z = np.load(os.path.join(HERE, "golden", "lowp_kernels.npz"))
DT = {"bf16": torch.bfloat16, "f16": torch.float16}
TABS = {"dopri5": tb.DOPRI5, "dopri8": tb.DOPRI8, "tsit5": tb.TSIT5, "bosh3": tb.BOSH3}


def T(a, dtype):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def ulps(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in units of the storage type's last place (both tensors of a 16-bit float type; finite values)."""
    ia, ib = a.view(torch.int16).to(torch.int32), b.view(torch.int16).to(torch.int32)
    ia = torch.where(ia < 0, -(ia & 0x7FFF), ia)
    ib = torch.where(ib < 0, -(ib & 0x7FFF), ib)
    return (ia - ib).abs()


def close_in_ulps(got, ref, max_ulp=1, max_frac=0.02):
    d = ulps(got, ref)
    assert int(d.max()) <= max_ulp, int(d.max())
    assert float((d > 0).float().mean()) <= max_frac, float((d > 0).float().mean())


@pytest.mark.parametrize("low", ["bf16", "f16"])
@pytest.mark.parametrize("method", list(TABS))
def test_step_pieces_equal_the_reference(method, low):
    dtype, tab, key = DT[low], TABS[method], f"{method}_{low}"
    y0 = T(Z[key + "_y0"], dtype)
    ks = list(T(Z[key + "_k"], dtype))
    dt = float(Z[key + "_t0_dt"][1])
    rtol, atol = (float(v) for v in Z[key + "_rtol_atol"])
    # stage inputs (rk_common.py:79): row i uses k_0..k_i
    rows = tab.beta_rows(False)
    for i, row in enumerate(rows):
        yi = olp.stage_combine(y0, [ks[j] for j in row.idx], row.coef, dt)
        close_in_ulps(yi, T(Z[key + "_stage_inputs"][i], dtype))
    y1 = T(Z[key + "_y1"], dtype)
    err = tb.SparseRow.from_dense(tab.c_error)
    e = olp._row_sum([ks[j] for j in err.idx], err.coef, dt)
    close_in_ulps(e, T(Z[key + "_y1_error"], dtype))
    # the quotient and the ratio from the REFERENCE's y1 (so that one differing last bit upstream is not amplified here)
    q = olp.error_quotient(y0, y1, [ks[j] for j in err.idx], err.coef, dt, rtol, atol)
    close_in_ulps(q, T(Z[key + "_quotient"], dtype), max_ulp=2, max_frac=0.03)
    # the norm exactly as the reference evaluates it, from the oracle's own terms
    sumsq, _ = olp.norm_terms(T(Z[key + "_quotient"], dtype))
    ref_ratio = float(Z[key + "_error_ratio"])
    # (ATen's mean: float32 sum / n, rounded once — a float16 sum of squares would overflow the type long before)
    mean = (torch.tensor(sumsq, dtype=torch.float64).to(torch.float32) / y0.numel()).to(dtype)
    assert abs(float(mean.sqrt()) - ref_ratio) <= 2 ** -7 * ref_ratio       # one rounding of the type
    # dense output
    mid = tb.SparseRow.from_dense(tab.c_mid)
    co = olp.quartic(y0, y1, ks[0], ks[-1], [ks[j] for j in mid.idx], mid.coef, dt)
    for plane, ref in zip(co, T(Z[key + "_interp_coeffs"], dtype)):
        d = ulps(plane, ref)
        # (y_mid enters with weights 16 / 32: one differing last bit of the row sum is a few units here)
        assert float((d > 0).float().mean()) <= 0.03 and int(d.max()) <= 64
    for x, ref in zip(Z[key + "_x_evals"], T(Z[key + "_y_evals"], dtype)):
        got = olp.dense_eval(y0, y1, ks[0], ks[-1], [ks[j] for j in mid.idx], mid.coef, dt, float(x))
        assert float((ulps(got, ref) > 0).float().mean()) <= 0.05


@pytest.mark.parametrize("low", ["bf16", "f16"])
def test_fixed_grid_pieces_equal_the_reference_bit_for_bit(low):
    """No row sums here: every operation is elementwise, so the oracle must reproduce the reference exactly."""
    dtype = DT[low]
    y0, ks, dt = T(Z[f"rk4_{low}_y0"], dtype), list(T(Z[f"rk4_{low}_k"], dtype)), float(Z[f"rk4_{low}_dt"])
    for stage in (1, 2, 3):
        got = olp.rk4_stage(stage, y0, ks[0], ks[1], ks[2], None, dt)
        assert torch.equal(got, T(Z[f"rk4_{low}_stage_inputs"][stage - 1], dtype)), stage
    assert torch.equal(olp.rk4_stage(4, y0, *ks, dt), T(Z[f"rk4_{low}_y1"], dtype))
    ya, yb = T(Z[f"lerp_{low}_ya_yb"], dtype)
    assert torch.equal(olp.lerp(ya, yb, float(Z[f"lerp_{low}_slope"])), T(Z[f"lerp_{low}_out"], dtype))
    yy, f0 = T(Z[f"init_{low}_y0_f0"], dtype)
    q0, q1 = olp.init_quotients(0, yy.reshape(-1), f0.reshape(-1), yy.reshape(-1), 1e-2, 1e-3)
    assert torch.equal(q0, T(Z[f"init_{low}_q0"], dtype).reshape(-1)) and torch.equal(q1, T(Z[f"init_{low}_q1"], dtype).reshape(-1))


_KernelOrderLow = _fallback.KernelOrderLowHostKernels


@pytest.mark.parametrize("low", ["bf16", "f16"])
def test_host_path_in_kernel_order_equals_the_oracle_bit_for_bit(low):
    """Every scalar-rounding decision of the host path (`_scalars.operand`, first vs second operands, pre-rounded
    coefficients) against the oracle's literal reference expressions — on the CPU, element for element."""
    dtype = DT[low]
    hk = _KernelOrderLow()
    g = torch.Generator().manual_seed(5)
    n = 4099
    r = lambda: torch.randn(n, generator=g, dtype=torch.float64).to(dtype)
    y0, y1, ks = r(), r(), [r() for _ in range(7)]
    coefs = (0.0371, -0.211, 0.5, 1.25, -0.0625, 0.33, 0.9)
    dt = 0.0371
    out = torch.empty_like(y0)
    for nt in (1, 2, 5, 7):
        hk.stage_combine(out, y0, ks[:nt], coefs[:nt], dt)
        assert torch.equal(out, olp.stage_combine(y0, ks[:nt], coefs[:nt], dt)), nt
    plan = hk.make_plan([(0, n, 1e-2, 1e-3)], n, 1024, torch.device("cpu"))
    scaled = torch.empty_like(y0)
    hk.error_norm(plan, y0, y1, ks[:6], coefs[:6], dt, scaled_out=scaled)
    q = olp.error_quotient(y0, y1, ks[:6], coefs[:6], dt, 1e-2, 1e-3)
    assert torch.equal(scaled, q)
    hk.init_scaled(plan, 0, ks[0], ks[1], y0, out, scaled)
    q0, q1 = olp.init_quotients(0, ks[0], ks[1], y0, 1e-2, 1e-3)
    assert torch.equal(out, q0) and torch.equal(scaled, q1)
    hk.init_scaled(plan, 1, ks[0], ks[1], y0, out)
    assert torch.equal(out, olp.init_quotients(1, ks[0], ks[1], y0, 1e-2, 1e-3)[0])
    for x in (0.0, 0.3, 1.0):
        hk.dense_eval(out, y0, y1, ks[0], ks[6], ks[:5], coefs[:5], dt, x)
        assert torch.equal(out, olp.dense_eval(y0, y1, ks[0], ks[6], ks[:5], coefs[:5], dt, x)), x
    planes = torch.empty(5, n, dtype=dtype)
    hk.interp_fit(planes, y0, y1, ks[0], ks[6], ks[:5], coefs[:5], dt)
    assert all(torch.equal(a, b) for a, b in zip(planes, olp.quartic(y0, y1, ks[0], ks[6], ks[:5], coefs[:5], dt)))
    for stage in (1, 2, 3, 4):
        hk.rk4_stage(stage, out, y0, ks[0], ks[1], ks[2], ks[3], 0.025)
        assert torch.equal(out, olp.rk4_stage(stage, y0, ks[0], ks[1], ks[2], ks[3], 0.025)), stage
    hk.lerp(out, y0, y1, 0.2417)
    assert torch.equal(out, olp.lerp(y0, y1, 0.2417))
    hk.fixed_stage(1, out, y0, ks[:1], (1 / 3,), 0.025)
    assert torch.equal(out, olp.fixed_stage(1, y0, ks[:1], (1 / 3,), 0.025))
    hk.fixed_stage(0, out, y0, ks[:3], (0.25, 0.1, 0.75), 0.025)
    assert torch.equal(out, olp.fixed_stage(0, y0, ks[:3], (0.25, 0.1, 0.75), 0.025))
    hk.weighted_sum(out, ks[:3], (0.7, -1.3, 2.0))
    assert torch.equal(out, olp.weighted_sum(ks[:3], (0.7, -1.3, 2.0)))
