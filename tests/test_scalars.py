"""Host scalar types (torchdiffeq_amd/_scalars.py) against ATen's own 0-dim arithmetic on the CPU: the rules the bf16 /
fp16 emulation and the initial-step heuristic rest on, checked on random operands (the measurement script
tools/lowfloat_semantics.py as a test), plus the complex-norm oracle against the torch-op host kernels."""
import math
import os
import random
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from torchdiffeq_amd import _fallback
from torchdiffeq_amd._scalars import (BFloat16Scalar, Float16Scalar, is_low, nextafter, operand, power, rdiv, real_dtype,
                                      scalar_type)


def test_low_scalars_round_like_aten_on_the_cpu():
    import lowfloat_semantics
    assert lowfloat_semantics.main(n=600) == 0


def test_scalar_type_table_and_promotion():
    assert scalar_type(torch.float32) is np.float32 and scalar_type(torch.complex128) is np.float64
    assert scalar_type(torch.bfloat16) is BFloat16Scalar and is_low(Float16Scalar) and not is_low(np.float32)
    assert real_dtype(torch.complex64) == torch.float32 and real_dtype(torch.bfloat16) == torch.bfloat16
    with pytest.raises(TypeError):
        scalar_type(torch.int32)
    b = BFloat16Scalar(0.3)
    # a wider 0-dim partner wins the promotion (0-dim bf16 op 0-dim fp64 -> fp64), a Python number never does
    assert type(b + np.float64(1.0)) is np.float64 and type(np.float32(2.0) * b) is np.float32
    assert type(b * 2.5) is BFloat16Scalar and type(1.0 - b) is BFloat16Scalar and type(-b) is BFloat16Scalar
    assert float(BFloat16Scalar(1.0) + 2.0 ** -9) == 1.0          # below half an ulp of bf16
    assert operand(np.float32, 0.1) == float(np.float32(0.1)) and operand(BFloat16Scalar, 0.1) == float(np.float32(0.1))
    assert float(Float16Scalar(1e-8)) == 0.0 and math.isinf(float(Float16Scalar(1e6)))


@pytest.mark.parametrize("cls,dtype", [(BFloat16Scalar, torch.bfloat16)])
def test_nextafter_matches_torch(cls, dtype):
    rng = random.Random(3)
    vals = [0.0, 1.0, -1.0, 2.0, 0.5, 255.0, 256.0, 3.3895e38, 1e-38, 9.2e-41] + \
           [rng.uniform(-1, 1) * 10 ** rng.uniform(-30, 30) for _ in range(500)]
    for v in vals:
        a = cls(v)
        for target in (a + 1, a - 1, cls(0.0), cls(math.inf)):
            want = float(torch.nextafter(torch.tensor(float(a), dtype=dtype), torch.tensor(float(target), dtype=dtype)))
            got = float(nextafter(a, target))
            assert got == want or (math.isnan(got) and math.isnan(want)), (v, float(target), got, want)
    assert float(nextafter(np.float32(1.0), np.float32(2.0))) == float(np.nextafter(np.float32(1.0), np.float32(2.0)))


def test_rdiv_and_power_are_atens_operations():
    rng = random.Random(5)
    for dtype, T in ((torch.float32, np.float32), (torch.float64, np.float64)):
        for _ in range(2000):
            x = abs(rng.uniform(-1, 1) * 10 ** rng.uniform(-4, 3))
            tx = torch.tensor(x, dtype=dtype)
            assert float(rdiv(0.01, T(x))) == float(0.01 / tx)
            for e in (0.2, 0.125, 0.5, 1.0 / 3.0):
                assert float(power(T(x), e)) == float(tx ** e)


@pytest.mark.parametrize("dtype", [torch.complex64, torch.complex128])
def test_complex_norm_oracle_equals_the_torch_op_host_kernels(dtype):
    """oracle/complex_norms.py (the checker of the complex HIP kernels) and `_fallback.KernelOrderHostKernels` (the
    kernels' left-to-right arithmetic in torch ops) are two restatements of misc.py:80-82 / 50-56 for complex states in
    the kernels' order; on the CPU they must agree bit for bit."""
    from oracle import complex_norms as cn
    g = torch.Generator().manual_seed(0)
    z = lambda n: torch.complex(torch.randn(n, generator=g, dtype=torch.float64), torch.randn(n, generator=g, dtype=torch.float64)).to(dtype)
    n, chunk = 300, 128
    segs = [(0, 100), (128, 72), (256, 44)]
    y0, y1, part = z(n), z(n), z(n)
    ks = [z(n) for _ in range(4)]
    coefs = [0.1, -0.25, 0.3, 0.05]
    hk = _fallback.KernelOrderHostKernels()
    plan = hk.make_plan([(off, m, 1e-3, 1e-6) for off, m in segs], n, chunk, "cpu")
    hk.error_norm(plan, y0, y1, ks, coefs, 0.05)
    r, _ = cn.error_ratio_parts(cn.error_estimate(ks, coefs, 0.05), y0, y1, 1e-3, 1e-6)
    assert plan.sums0 == cn.segment_sums(r, segs)
    hk.error_norm_partial(plan, part, y0, y1, ks[:2], coefs[:2], 0.05)
    r, _ = cn.error_ratio_parts(cn.error_estimate(ks[:2], coefs[:2], 0.05, partial=part), y0, y1, 1e-3, 1e-6)
    assert plan.sums0 == cn.segment_sums(r, segs)
    for mode in (0, 1):
        hk.init_norms(plan, mode, ks[0], ks[1], y0)
        q0, q1 = cn.init_quotients(mode, ks[0], ks[1], y0, 1e-3, 1e-6)
        assert plan.sums0 == cn.segment_sums(q0, segs)
        if mode == 0:
            assert plan.sums1 == cn.segment_sums(q1, segs)
