"""BASELINE.json's configurations at FULL size on the MI355X, checked through size-independent
properties (closed forms, round trips, additivity of batch-summed gradients) — the CPU oracle cannot
finish these sizes in seconds, so the oracle / golden comparisons live in the reduced-size tests."""
import math

import pytest
import torch

import torchdiffeq_amd as tda
from _cases import StatFunc, rel_err

pytestmark = pytest.mark.gpu


def _linear(B, D, dtype):
    g = torch.Generator().manual_seed(0)
    G = torch.randn(D, D, generator=g, dtype=torch.float64) / D ** 0.5
    A = 0.5 * (G - G.T) - 0.1 * torch.eye(D, dtype=torch.float64)
    y0 = torch.randn(B, D, generator=g, dtype=torch.float64)
    return A.to(dtype).cuda(), y0.to(dtype).cuda()


def test_cfg2_full_size_closed_form_and_step_counts():
    """cfg2: dopri5, 65536 x 128 fp32, reference-default tolerances.  Reference on CPU: NFE 68 =
    2 + 11*6, 11 accepted / 0 rejected, 2.4e-6 from the closed form."""
    A, y0 = _linear(65536, 128, torch.float32)
    At = A.T.contiguous()
    f = StatFunc(lambda t, y: y @ At)
    with torch.no_grad():
        y = tda.odeint(f, y0, torch.tensor([0.0, 1.0]).cuda(), method="dopri5")
    exact = y0.double() @ torch.linalg.matrix_exp(A.double()).T
    assert rel_err(y[-1], exact) < 1e-5
    assert abs(len(f.accept) - 11) <= 1 and len(f.reject) <= 1
    assert f.nfe == 2 + 6 * (len(f.accept) + len(f.reject))


def test_cfg4_full_size_dopri8_fp64():
    """cfg4: dopri8, 16384 x 512 fp64, rtol 1e-9.  Reference on CPU: NFE 67, 5 accepted, 4.0e-7 from expm."""
    A, y0 = _linear(16384, 512, torch.float64)
    At = A.T.contiguous()
    f = StatFunc(lambda t, y: y @ At)
    with torch.no_grad():
        y = tda.odeint(f, y0, torch.tensor([0.0, 1.0], dtype=torch.float64).cuda(), method="dopri8", rtol=1e-9, atol=1e-11)
    exact = y0 @ torch.linalg.matrix_exp(A).T
    assert rel_err(y[-1], exact) < 2e-6
    assert abs(len(f.accept) - 5) <= 1 and len(f.reject) <= 1
    assert f.nfe == 2 + 13 * (len(f.accept) + len(f.reject))


def test_cfg2_full_size_per_element_tolerances_equal_scalar_ones():
    """cfg2 with the reference-default rtol as a CONSTANT per-element vector (ABI 20: tdeq_error_norm_vec_ctrl continuing the
    partial error row, carried partial sums, look-ahead): the same steps as the scalar-tolerance solve — only the quotient's
    precision differs (fp64 when a tolerance is dimensioned, misc.py:80-82) — and a row-wise vector that is tighter on half of
    the columns costs more steps and lands closer to the closed form there."""
    A, y0 = _linear(65536, 128, torch.float32)
    At = A.T.contiguous()
    t = torch.tensor([0.0, 1.0]).cuda()
    exact = (y0.double() @ torch.linalg.matrix_exp(A.double()).T)
    runs = {}
    for name, rtol in (("scalar", 1e-7), ("vector", torch.full((65536, 128), 1e-7, dtype=torch.float64, device="cuda")),
                       ("tight_half", torch.cat([torch.full((64,), 1e-9, dtype=torch.float64),
                                                 torch.full((64,), 1e-7, dtype=torch.float64)]).cuda())):
        f = StatFunc(lambda t_, y: y @ At)
        with torch.no_grad():
            y = tda.odeint(f, y0, t, method="dopri5", rtol=rtol, atol=1e-9)[-1]
        runs[name] = (y, f.nfe, len(f.accept), len(f.reject))
    assert runs["vector"][1:] == runs["scalar"][1:]
    assert rel_err(runs["vector"][0], runs["scalar"][0]) < 1e-5      # (fp32 rounding level: a step size that differs in its last bits)
    assert runs["tight_half"][1] > runs["scalar"][1]
    err = lambda y: float((y.double() - exact).abs().max() / exact.abs().max())
    assert err(runs["tight_half"][0]) <= err(runs["scalar"][0]) * 1.05 and err(runs["scalar"][0]) < 1e-5


def test_cfg2_full_size_round_trip():
    """0 -> 1 -> 0 returns to y0 (exercises the folded time reversal at full size)."""
    A, y0 = _linear(65536, 128, torch.float32)
    At = A.T.contiguous()
    f = lambda t, y: y @ At
    with torch.no_grad():
        y1 = tda.odeint(f, y0, torch.tensor([0.0, 1.0]).cuda(), rtol=1e-6, atol=1e-8)[-1]
        yb = tda.odeint(f, y1, torch.tensor([1.0, 0.0]).cuda(), rtol=1e-6, atol=1e-8)[-1]
    assert rel_err(yb, y0) < 2e-5


def _mlp():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.Tanh(), torch.nn.Linear(256, 256), torch.nn.Tanh(),
                              torch.nn.Linear(256, 64)).cuda()

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = net

        def forward(self, t, y):
            return self.net(y)

    return F()


def test_cfg3_full_size_adjoint_gradient_additivity():
    """cfg3: odeint_adjoint, MLP 64-256-256-64, 65536 x 64 fp32, rtol 1e-5.  The parameter gradient is a
    sum over batch rows, so grads(full batch) == grads(first half) + grads(second half) up to the solve
    tolerance (each solve picks its own steps) — the property the 8-GPU all-reduce relies on; dL/dy0 of
    a row does not depend on which shard it is solved in."""
    f = _mlp()
    g = torch.Generator().manual_seed(1)
    y0 = torch.randn(65536, 64, generator=g).cuda()
    t = torch.tensor([0.0, 1.0]).cuda()

    def grads(rows):
        for p in f.parameters():
            p.grad = None
        x = y0[rows].clone().requires_grad_(True)
        y = tda.odeint_adjoint(f, x, t, rtol=1e-5, atol=1e-7, method="dopri5")
        y[-1].pow(2).sum().backward()
        return x.grad, [p.grad.clone() for p in f.parameters()]

    gy_full, gp_full = grads(slice(0, 65536))
    gy_a, gp_a = grads(slice(0, 32768))
    gy_b, gp_b = grads(slice(32768, 65536))
    assert rel_err(torch.cat([gy_a, gy_b]), gy_full) < 2e-4
    for full, a, b in zip(gp_full, gp_a, gp_b):
        assert rel_err(a + b, full) < 2e-4
    assert all(torch.isfinite(p).all() for p in gp_full)


def test_cfg3_shape_adjoint_matches_backprop_through_unrolled_rk4():
    """Gradient correctness at cfg3's layer sizes: adjoint (dopri5, tight) vs autograd through a
    hand-unrolled fixed-step RK4 of the same MLP field on a batch slice (independent of our solver)."""
    f = _mlp().double()
    g = torch.Generator().manual_seed(2)
    y0 = torch.randn(512, 64, generator=g, dtype=torch.float64).cuda().requires_grad_(True)
    t = torch.tensor([0.0, 1.0], dtype=torch.float64).cuda()
    y = tda.odeint_adjoint(f, y0, t, rtol=1e-9, atol=1e-11, method="dopri5")
    y[-1].pow(2).sum().backward()
    g_adj_y = y0.grad.clone()
    g_adj_p = [p.grad.clone() for p in f.parameters()]
    y0.grad = None
    for p in f.parameters():
        p.grad = None
    n, h, yy = 200, 1.0 / 200, y0
    for i in range(n):
        k1 = f(None, yy)
        k2 = f(None, yy + 0.5 * h * k1)
        k3 = f(None, yy + 0.5 * h * k2)
        k4 = f(None, yy + h * k3)
        yy = yy + (h / 6) * (k1 + 2 * k2 + 2 * k3 + k4)
    yy.pow(2).sum().backward()
    assert rel_err(g_adj_y, y0.grad) < 1e-6
    for a, p in zip(g_adj_p, f.parameters()):
        assert rel_err(a, p.grad) < 1e-6


def test_kernels_beyond_2_31_elements(hip_kernels):
    """One state tensor of 2^31 + 2^20 + 3 fp32 elements (8.6 GB; the MI355X holds 288 GB): 64-bit indexing of the
    streaming kernels (grid-stride beyond 65,536 workgroups) and of the norm kernels (> 2^20 reduction chunks).
    Checked on windows at the start, across the 2^31 boundary and at the unaligned tail against the same rounding
    sequence in torch; the error sums against an fp64 torch reduction; a planted inf is counted once."""
    import numpy as np
    from torchdiffeq_amd import _native
    free, _ = torch.cuda.mem_get_info()
    n = (1 << 31) + (1 << 20) + 3
    if free < 7 * n * 4:
        pytest.skip("not enough free HBM for the 2^31-element case")
    dev = torch.device("cuda:0")

    def ramp(scale, shift):          # cheap deterministic data, different at every index window
        x = torch.arange(n, device=dev, dtype=torch.int64)
        return (((x * 2654435761 + shift) % 1000003).to(torch.float32) / 1000003.0 - 0.5) * scale

    y0, k0, k1 = ramp(2.0, 1), ramp(1.0, 7), ramp(3.0, 13)
    out = torch.empty(n, device=dev)
    dt, coefs = 0.0371, [0.3, -1.7]
    hip_kernels.stage_combine(out, y0, [k0, k1], coefs, dt)
    c = [np.float32(np.float32(cf) * np.float32(dt)) for cf in coefs]
    windows = [slice(0, 4096), slice((1 << 31) - 2048, (1 << 31) + 2048), slice(n - 4099, n)]
    for w in windows:
        ref = y0[w] + (k0[w] * float(c[0]) + k1[w] * float(c[1]))
        assert torch.equal(out[w], ref), w
    # fused end-of-step pair + norm over > 2^20 chunks
    y1, epart = torch.empty(n, device=dev), torch.empty(n, device=dev)
    hip_kernels.stage_combine_err(y1, epart, y0, [k0, k1], coefs, [1e-3, -2e-3], dt)
    for w in windows:
        assert torch.equal(y1[w], out[w])
    del out
    chunk = _native.pick_chunk(n)
    plan = hip_kernels.make_plan([(0, n, 1e-5, 1e-7)], n, chunk, dev)
    assert plan.n_chunks > (1 << 20)
    hip_kernels.error_norm_partial(plan, epart, y0, y1, [k0], [5e-4], dt)
    sumsq, _, bad = hip_kernels.read_norms(plan)
    assert bad == [0.0]
    ce = np.float32(np.float32(5e-4) * np.float32(dt))
    total = 0.0
    step = 1 << 27
    for lo in range(0, n, step):
        w = slice(lo, min(lo + step, n))
        e = epart[w] + k0[w] * float(ce)
        tol = 1e-7 + 1e-5 * torch.maximum(y0[w].abs(), y1[w].abs())
        r = e / tol.to(torch.float32)
        total += float((r.double() ** 2).sum())
    assert sumsq[0] == pytest.approx(total, rel=1e-6)
    y1[(1 << 31) + 5] = float("inf")
    hip_kernels.error_norm_partial(plan, epart, y0, y1, [k0], [5e-4], dt)
    _, _, bad = hip_kernels.read_norms(plan)
    assert bad == [1.0]


def test_adams_full_size_closed_form_and_linearity():
    """The Adams methods at the cfg2 state (65536 x 128 fp32): closed form, and linearity in y0 for a linear field —
    solve(a*y0) == a*solve(y0) to rounding, a size-independent property of every step formula."""
    A, y0 = _linear(65536, 128, torch.float32)
    At = A.T.contiguous()
    f = lambda t, y: y @ At
    t = torch.tensor([0.0, 1.0]).cuda()
    exact = y0.double() @ torch.linalg.matrix_exp(A.double()).T
    with torch.no_grad():
        for method, kw, tol in [("explicit_adams", dict(options=dict(step_size=1 / 64, max_order=6)), 1e-5),
                                ("implicit_adams", dict(options=dict(step_size=1 / 32), rtol=1e-5, atol=1e-7), 1e-5)]:
            y = tda.odeint(f, y0, t, method=method, **kw)[-1]
            assert rel_err(y, exact) < tol, method
            y2 = tda.odeint(f, y0 * 2.0, t, method=method, **kw)[-1]
            assert rel_err(y2, y * 2.0) < 1e-6, method       # scaling by 2 is exact in fp32: only the census can differ


def test_implicit_rk_full_size_matrix_free():
    """gl4 at the cfg2 state: 2 x 8.4M unknowns per step — the reference's dense Broyden matrix would have 2.8e14
    entries.  Closed form; the opt-in RMS residual test keeps the iteration count at a handful."""
    A, y0 = _linear(65536, 128, torch.float32)
    At = A.T.contiguous()
    calls = [0]

    def f(t, y):
        calls[0] += 1
        return y @ At
    with torch.no_grad():
        y = tda.odeint(f, y0, torch.tensor([0.0, 1.0]).cuda(), method="gl4",
                       options=dict(step_size=0.125, residual_norm="rms"))[-1]
    exact = y0.double() @ torch.linalg.matrix_exp(A.double()).T
    assert rel_err(y, exact) < 1e-5
    assert calls[0] < 8 * (1 + 2 * 12)          # a handful of Broyden iterations per step, not max_iters = 100


@pytest.mark.parametrize("method", ["dopri5", "rk4"])
def test_bf16_state_at_full_size_equals_the_torch_twin_bit_for_bit(method):
    """r05, BASELINE.json's size with a bfloat16 STATE (65536 x 128 = 8.4 M elements, 16-byte lane accesses of 8 elements,
    chunk 2048, the device controller and the look-ahead stage in bf16): the HIP kernels of csrc/tdeq_kernels_lp.hpp against
    the same arithmetic in torch ops on the same device (`_fallback.KernelOrderLowHostKernels`, host-driven) — a
    size-independent property: every output row bit-identical, equal evaluation counts."""
    from torchdiffeq_amd import _fallback, _native
    A, y0 = _linear(65536, 128, torch.float32)
    A = (A + 0.1 * torch.eye(128, device=A.device)).to(torch.bfloat16)      # the skew-symmetric part: |y| stays O(1)
    y0 = y0.to(torch.bfloat16)
    At = A.T.contiguous()
    t = torch.linspace(0.0, 2.0, 4, device="cuda")
    nfe = [0]

    def f(tt, y):
        nfe[0] += 1
        return y @ At
    kw = dict(rtol=1e-2, atol=1e-3) if method == "dopri5" else dict(options=dict(step_size=0.125))
    with torch.no_grad():
        y_hip = tda.odeint(f, y0, t, method=method, **kw)
    n_hip, nfe[0] = nfe[0], 0
    twin = _fallback.KernelOrderLowHostKernels()
    orig = _native.get_kernels
    _native.get_kernels = lambda device, dtype=None: twin if dtype == torch.bfloat16 else orig(device, dtype)
    try:
        with torch.no_grad():
            y_twin = tda.odeint(f, y0, t, method=method, **kw)
    finally:
        _native.get_kernels = orig
    assert n_hip == nfe[0] and n_hip > 10
    assert y_hip.dtype == torch.bfloat16 and torch.isfinite(y_hip.float()).all()
    assert torch.equal(y_hip.view(torch.int16), y_twin.view(torch.int16))
