"""complex64 / complex128 states on the HIP kernels (r04): linear operations through the real kernels on the (re, im)
view, tolerance-scaled norms through csrc/tdeq_kernels_complex.hpp.  Checked (i) kernel by kernel against the reference's
own expressions (oracle/complex_norms.py <- torchdiffeq/_impl/misc.py:80-82, 50-56) evaluated with ATen on the SAME
device — element for element bit-exact, sums to 1e-12 — and on the CPU oracle (last-bit libm differences: 1e-6 / 1e-14);
(ii) whole solves against the package's torch-op path forced onto the same device (bit-identical, equal evaluation
counts) and against the reference's outputs (tests/golden/hostpath.npz)."""
import warnings

import numpy as np
import pytest
import torch

import torchdiffeq_amd as tda
from torchdiffeq_amd import _fallback, _native
from oracle import complex_norms as cn
from _cases import T, load

pytestmark = pytest.mark.gpu
DEV = "cuda"
CDT = {"c64": torch.complex64, "c128": torch.complex128}


def _z(n, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    re = torch.randn(n, generator=g, dtype=torch.float64) * scale
    im = torch.randn(n, generator=g, dtype=torch.float64) * scale
    return torch.complex(re, im).to(dtype).to(DEV)


def _kern(dtype):
    k = _native.get_kernels(torch.device("cuda:0"), dtype)
    assert isinstance(k, _native.ComplexHipKernels)
    return k


def _layout(n, chunk, n_seg):
    """n_seg segments of roughly equal size with chunk-aligned starts (element offsets), total padded length."""
    per = n // n_seg
    segs, off = [], 0
    for s in range(n_seg):
        m = per if s + 1 < n_seg else n - per * (n_seg - 1)
        segs.append((off, m))
        off += -(-m // chunk) * chunk
    return segs, off


@pytest.mark.parametrize("tag", ["c64", "c128"])
@pytest.mark.parametrize("n,chunk,n_seg,offset,nt", [(1000, 1024, 1, 0, 1), (4099, 1024, 1, 0, 6), (4099, 1024, 1, 1, 6),
                                                     (4099, 1024, 1, 1, 9), (1 << 20, 2048, 1, 0, 6), (300000, 1024, 3, 0, 9),
                                                     (5000, 1024, 20, 0, 1), (5000, 1024, 20, 0, 9)])
def test_complex_error_norm_kernel(tag, n, chunk, n_seg, offset, nt):
    dtype = CDT[tag]
    kern = _kern(dtype)
    segs, total = _layout(n, chunk, n_seg)
    mk = lambda seed, scale=1.0: _z(total + offset, dtype, seed, scale)[offset:]      # offset 1: not 16-byte aligned (c64)
    y0, y1 = mk(1), mk(2)
    ks = [mk(10 + j) for j in range(nt)]
    coefs = [0.11 * (-1) ** j * (j + 1) for j in range(nt)]
    dt, rtol, atol = 0.037, 1e-4, 1e-6
    plan = kern.make_plan([(off, m, rtol, atol) for off, m in segs], total, chunk, torch.device(DEV))
    scaled = torch.full_like(y0, complex(7, 7))
    kern.error_scaled(plan, scaled, y0, y1, ks, coefs, dt)
    sums, _, bad = kern.read_norms(plan)
    kern.error_norm(plan, y0, y1, ks, coefs, dt)
    sums2, _, _ = kern.read_norms(plan)
    assert sums == sums2
    # the same expressions with ATen on the same device: bit-exact elementwise
    err = cn.error_estimate(ks, coefs, dt)
    r, _ = cn.error_ratio_parts(err, y0, y1, rtol, atol)
    for off, m in segs:
        assert torch.equal(torch.view_as_real(scaled[off:off + m]), torch.view_as_real(r[off:off + m]))
    if n_seg > 1:       # padding of a segmented layout is zero-filled
        mask = torch.ones(total, dtype=torch.bool, device=DEV)
        for off, m in segs:
            mask[off:off + m] = False
        assert float(scaled[mask].abs().max()) == 0.0
    ref = cn.segment_sums(r, segs)
    np.testing.assert_allclose(sums, ref, rtol=1e-12)
    assert all(b == 0 for b in bad)
    # ... and on the CPU oracle (glibc hypot vs the device's: last-bit differences of |z|)
    r_cpu, _ = cn.error_ratio_parts(cn.error_estimate([k.cpu() for k in ks], coefs, dt), y0.cpu(), y1.cpu(), rtol, atol)
    np.testing.assert_allclose(sums, cn.segment_sums(r_cpu, segs), rtol=1e-5 if tag == "c64" else 1e-13)
    # non-finite census counts complex ELEMENTS
    y0b = y0.clone()
    y0b[segs[-1][0] + 3] = complex(float("nan"), 1.0)
    y0b[segs[0][0]] = complex(0.0, float("inf"))
    kern.error_norm(plan, y0b, y1, ks, coefs, dt)
    _, _, bad = kern.read_norms(plan)
    assert sum(bad) == 2 and bad[0] >= 1 and bad[-1] >= 1


@pytest.mark.parametrize("tag", ["c64", "c128"])
@pytest.mark.parametrize("n,chunk,n_seg,nt", [(4099, 1024, 1, 0), (4099, 1024, 1, 2), (1 << 20, 2048, 1, 1), (300000, 1024, 3, 2)])
def test_complex_error_norm_partial_kernel(tag, n, chunk, n_seg, nt):
    dtype = CDT[tag]
    kern = _kern(dtype)
    segs, total = _layout(n, chunk, n_seg)
    y0, y1, part = _z(total, dtype, 1), _z(total, dtype, 2), _z(total, dtype, 3, 1e-3)
    ks = [_z(total, dtype, 10 + j) for j in range(nt)]
    coefs = [0.013, -0.021][:nt]
    dt, rtol, atol = 0.05, 1e-5, 1e-7
    plan = kern.make_plan([(off, m, rtol, atol) for off, m in segs], total, chunk, torch.device(DEV))
    kern.error_norm_partial(plan, part, y0, y1, ks, coefs, dt)
    sums, _, bad = kern.read_norms(plan)
    err = cn.error_estimate(ks, coefs, dt, partial=part) if nt else part
    r, _ = cn.error_ratio_parts(err, y0, y1, rtol, atol)
    np.testing.assert_allclose(sums, cn.segment_sums(r, segs), rtol=1e-12)
    assert all(b == 0 for b in bad)


@pytest.mark.parametrize("tag", ["c64", "c128"])
@pytest.mark.parametrize("n,chunk,n_seg", [(4099, 1024, 1), (1 << 20, 2048, 1), (300000, 1024, 3)])
def test_complex_init_norms_kernels(tag, n, chunk, n_seg):
    dtype = CDT[tag]
    kern = _kern(dtype)
    segs, total = _layout(n, chunk, n_seg)
    a, b, y = _z(total, dtype, 1), _z(total, dtype, 2), _z(total, dtype, 3)
    rtol, atol = 1e-3, 1e-5
    plan = kern.make_plan([(off, m, rtol, atol) for off, m in segs], total, chunk, torch.device(DEV))
    for mode in (0, 1):
        q0, q1 = cn.init_quotients(mode, a, b, y, rtol, atol)
        kern.init_norms(plan, mode, a, b, y)
        s0, s1, bad = kern.read_norms(plan)
        np.testing.assert_allclose(s0, cn.segment_sums(q0, segs), rtol=1e-12)
        if mode == 0:
            np.testing.assert_allclose(s1, cn.segment_sums(q1, segs), rtol=1e-12)
        assert all(v == 0 for v in bad)
        o0, o1 = torch.full_like(a, 5), torch.full_like(a, 5)
        kern.init_scaled(plan, mode, a, b, y, o0, o1 if mode == 0 else None)
        for off, m in segs:
            assert torch.equal(torch.view_as_real(o0[off:off + m]), torch.view_as_real(q0[off:off + m]))
            if mode == 0:
                assert torch.equal(torch.view_as_real(o1[off:off + m]), torch.view_as_real(q1[off:off + m]))


@pytest.mark.parametrize("tag", ["c64", "c128"])
def test_linear_operations_on_the_real_view_equal_the_torch_op_path(tag):
    """stage combine / dense output / rk4 / lerp / pack through ComplexHipKernels vs `_fallback.HostKernels` (torch ops on
    complex tensors) on the same device: bit-identical."""
    dtype = CDT[tag]
    kern, host = _kern(dtype), _fallback.KernelOrderHostKernels()
    n = 5000
    y0, y1 = _z(n, dtype, 1), _z(n, dtype, 2)
    ks = [_z(n, dtype, 10 + j) for j in range(7)]
    coefs = [0.2, -0.1, 0.3, 0.05, -0.4, 0.15, 0.01]
    eq = lambda a, b: torch.equal(torch.view_as_real(a), torch.view_as_real(b))
    for nt in (1, 3, 7):
        a, b = torch.empty_like(y0), torch.empty_like(y0)
        kern.stage_combine(a, y0, ks[:nt], coefs[:nt], 0.07)
        host.stage_combine(b, y0, ks[:nt], coefs[:nt], 0.07)
        assert eq(a, b)
    a, b, ea, eb = (torch.empty_like(y0) for _ in range(4))
    kern.stage_combine_err(a, ea, y0, ks[:6], coefs[:6], coefs[1:7], -0.03)
    host.stage_combine_err(b, eb, y0, ks[:6], coefs[:6], coefs[1:7], -0.03)
    assert eq(a, b) and eq(ea, eb)
    rows_a, rows_b = torch.empty(3, n, dtype=dtype, device=DEV), torch.empty(3, n, dtype=dtype, device=DEV)
    kern.dense_eval_multi(rows_a, y0, y1, ks[0], ks[6], ks[:6], coefs[:6], 0.1, [0.2, 0.5, 0.9])
    host.dense_eval_multi(rows_b, y0, y1, ks[0], ks[6], ks[:6], coefs[:6], 0.1, [0.2, 0.5, 0.9])
    assert eq(rows_a, rows_b)
    for stage in (1, 2, 3, 4):
        kern.rk4_stage(stage, a, y0, ks[0], ks[1], ks[2], ks[3], 0.05)
        host.rk4_stage(stage, b, y0, ks[0], ks[1], ks[2], ks[3], 0.05)
        assert eq(a, b)
    kern.lerp(a, y0, y1, 0.3)
    host.lerp(b, y0, y1, 0.3)
    assert eq(a, b)
    # real-part dot products of the autograd nodes
    d = kern.multi_dot(y0, ks[:3])
    np.testing.assert_allclose(d.cpu().numpy(), host.multi_dot(y0, ks[:3]).cpu().numpy(), rtol=1e-12 if tag == "c128" else 1e-6)


CASES = [(tag, method, d) for tag in ("c64", "c128") for method in ("dopri5", "dopri8", "rk4", "bosh3", "tsit5") for d in ("fwd", "rev")]


def _solve(z, tag, method, d, **extra):
    A, y0 = T(z[f"{tag}_A"], DEV), T(z[f"{tag}_y0"], DEV)
    t = T(z[f"{tag}_dopri5_{d}_t"], DEV)
    kw = {"dopri5": dict(rtol=1e-5, atol=1e-7), "dopri8": dict(rtol=1e-6, atol=1e-8), "tsit5": dict(rtol=1e-5, atol=1e-7),
          "rk4": dict(options=dict(step_size=0.05)), "bosh3": dict(rtol=1e-4, atol=1e-6)}[method]
    kw = dict(kw)
    if extra:
        kw["options"] = dict(kw.get("options", {}), **extra)
    nfe = [0]

    def f(t_, y_):
        assert not t_.is_complex()
        nfe[0] += 1
        return y_ @ A.T
    with torch.no_grad():
        y = tda.odeint(f, y0, t, method=method, **kw)
    return y, nfe[0]


@pytest.mark.parametrize("tag,method,d", CASES)
def test_complex_solves_on_the_kernels_equal_the_torch_op_path_and_the_reference(monkeypatch, tag, method, d):
    z = load("hostpath.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("error", _fallback.HostPathWarning)          # a complex cuda state must not take the host path
        y, nfe = _solve(z, tag, method, d)
    assert y.is_cuda and y.dtype == CDT[tag]
    if method != "tsit5":
        ref = z[f"{tag}_{method}_{d}_y"]
        err = float((y.cpu() - T(ref)).abs().max() / np.abs(ref).max())
        assert err < (2e-5 if tag == "c64" else (1e-7 if method == "dopri8" else 1e-9)), err
        if method != "rk4":     # (rk4: the solver's arithmetic is the reference's bit for bit, func's complex GEMM is the device's)
            assert nfe == int(z[f"{tag}_{method}_{d}_nfe"])
    # the torch-op path on the same device (r03's route for complex states)
    host = _fallback.KernelOrderHostKernels()
    monkeypatch.setattr(_native, "get_kernels", lambda device, dtype=None: host)
    y_host, nfe_host = _solve(z, tag, method, d)
    monkeypatch.undo()
    assert nfe_host == nfe
    if method in ("rk4",):
        assert torch.equal(torch.view_as_real(y), torch.view_as_real(y_host))
    else:
        # identical arithmetic per element; the norm sums are accumulated in a different order (workgroup tree vs
        # torch.sum), so a step size may differ in its last bit
        err = float((y - y_host).abs().max() / y_host.abs().max())
        assert err < (5e-6 if tag == "c64" else 1e-13), err


@pytest.mark.parametrize("tag", ["c64", "c128"])
def test_complex_solve_with_captured_steps_and_lookahead(tag):
    """The device-resident controller, the look-ahead stage and hipGraph replays work on complex states too (the
    controller sees sums, the stage kernels the real view): same solution as the host-driven loop."""
    z = load("hostpath.npz")
    y_plain, nfe = _solve(z, tag, "dopri5", "fwd")
    y_graph, _ = _solve(z, tag, "dopri5", "fwd", hip_graph=True)
    import os
    os.environ["TDEQ_LOOKAHEAD"] = "0"
    try:
        y_host, nfe_host = _solve(z, tag, "dopri5", "fwd")
    finally:
        os.environ.pop("TDEQ_LOOKAHEAD")
    assert nfe == nfe_host
    # host-driven loop: the controller's `pow` is libm's instead of the device's (<= 2 ulp of fp64 in dt_next, docs/LAB_NOTEBOOK.md §8)
    err = float((y_plain - y_host).abs().max() / y_host.abs().max())
    assert err < (1e-5 if tag == "c64" else 1e-12), err
    # captured steps use the same device controller as the look-ahead path: the same bits
    assert torch.equal(torch.view_as_real(y_plain), torch.view_as_real(y_graph))


def test_gradients_through_a_complex_solve_on_the_kernels():
    z = load("hostpath.npz")
    A = T(z["c128_A"], DEV)
    y0 = T(z["c128_y0"], DEV).requires_grad_(True)
    y = tda.odeint(lambda t_, y_: y_ @ A.T, y0, torch.tensor([0.0, 1.0], dtype=torch.float64, device=DEV), method="dopri5",
                   rtol=1e-7, atol=1e-9)
    (y[-1].abs() ** 2).sum().backward()
    ref = T(z["c128_grad_y0"])
    assert float((y0.grad.cpu() - ref).abs().max() / ref.abs().max()) < 1e-8
    # adjoint, tuple state with a complex and a real component
    lin = torch.nn.Linear(4, 4, bias=False).to(DEV).to(torch.complex128)
    a0 = torch.randn(3, 4, dtype=torch.complex128, device=DEV)
    grads = []
    for fn in (tda.odeint_adjoint, tda.odeint):
        lin.zero_grad()
        x = a0.clone().requires_grad_(True)
        ya = fn(lambda t_, s: lin(s) * 0.3, x, torch.tensor([0.0, 0.7], dtype=torch.float64, device=DEV), rtol=1e-9,
                atol=1e-11, method="dopri5", **({"adjoint_params": tuple(lin.parameters())} if fn is tda.odeint_adjoint else {}))
        ya[-1].abs().pow(2).sum().backward()
        grads.append((x.grad.clone(), lin.weight.grad.clone()))
    assert float((grads[0][0] - grads[1][0]).abs().max() / grads[1][0].abs().max()) < 1e-6
    assert float((grads[0][1] - grads[1][1]).abs().max() / grads[1][1].abs().max()) < 1e-6
