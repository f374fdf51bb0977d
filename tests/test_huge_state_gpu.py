"""Maximum sizes: a flat state of MORE than 2^31 elements (fp32: 8.6 GB per state-sized tensor — 3 % of the MI355X's 288 GB;
the layouts are sized for that HBM, docs/LAB_NOTEBOOK.md §2).  Element indices, the exact-cover grid and the chunk table must be
64-bit throughout: stage combine, fused error combine, the segmented error norm (2^20 + 3 chunks of partials) and a whole
adaptive solve, checked at windows around 0, 2^31 and the end and against chunked torch reductions."""
import math

import pytest
import torch

import torchdiffeq_amd as tda

pytestmark = pytest.mark.gpu
N = (1 << 31) + 4099
STEP = 1 << 27


@pytest.fixture(autouse=True)
def _release_cached_blocks():
    yield
    torch.cuda.empty_cache()          # hand the ~100 GB back: later tests start subprocesses on the same GPU


def _need_memory(gb):
    free, _ = torch.cuda.mem_get_info()
    if free < gb * (1 << 30):
        pytest.skip(f"needs {gb} GB of free device memory")


def _pattern(n, mod, scale, shift=0.0):
    """x[i] = ((i * 7 + shift) mod `mod`) * scale, filled in slices (no 17 GB int64 temporary)."""
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    for lo in range(0, n, STEP):
        hi = min(n, lo + STEP)
        i = torch.arange(lo, hi, device="cuda", dtype=torch.int64)
        out[lo:hi] = ((i * 7 + int(shift)) % mod).to(torch.float32) * scale
        del i
    return out


def _windows(n):
    return [slice(0, 4096), slice((1 << 31) - 2048, (1 << 31) + 2048), slice(n - 4099, n)]


def test_combine_and_norm_beyond_2_31_elements(hip_kernels):
    _need_memory(60)
    k = hip_kernels
    y0 = _pattern(N, 1021, 1e-3)
    k0 = _pattern(N, 509, -2e-3, shift=3)
    out = torch.empty(N, dtype=torch.float32, device="cuda")
    err = torch.empty(N, dtype=torch.float32, device="cuda")
    dt, c, ce = 0.125, 0.5, 0.25
    k.stage_combine_err(out, err, y0, [k0], [c], [ce], dt)
    torch.cuda.synchronize()
    cT, ceT = torch.tensor(c * dt, dtype=torch.float32), torch.tensor(ce * dt, dtype=torch.float32)
    for w in _windows(N):
        assert torch.equal(out[w], y0[w] + k0[w] * cT)
        assert torch.equal(err[w], k0[w] * ceT)
    plain = torch.empty(N, dtype=torch.float32, device="cuda")
    k.stage_combine(plain, y0, [k0], [c], dt)
    torch.cuda.synchronize()
    for lo in range(0, N, STEP):                 # the whole vector, slice by slice
        hi = min(N, lo + STEP)
        assert torch.equal(plain[lo:hi], out[lo:hi])
    del plain
    # segmented error norm over two segments whose boundary lies beyond 2^31 elements
    chunk = 2048
    split = ((1 << 31) // chunk + 1) * chunk
    rtol, atol = 1e-3, 1e-4
    plan = k.make_plan([(0, split, rtol, atol), (split, N - split, 2 * rtol, atol)], N, chunk, torch.device("cuda:0"))
    k.error_norm(plan, y0, out, [k0], [ce], dt)
    sums, _, bad = k.read_norms(plan)
    assert bad == [0.0, 0.0]
    want = [0.0, 0.0]
    for lo in range(0, N, STEP):
        hi = min(N, lo + STEP)
        for s, (a, b, rt) in enumerate(((0, split, rtol), (split, N, 2 * rtol))):
            a2, b2 = max(a, lo), min(b, hi)
            if a2 < b2:
                e = (k0[a2:b2] * ceT).double()
                tol = atol + rt * torch.maximum(y0[a2:b2].abs(), out[a2:b2].abs()).double()
                want[s] += float(((e / tol) ** 2).sum())
    for got, ref in zip(sums, want):
        assert got == pytest.approx(ref, rel=1e-5)          # tolerance products are formed in fp32 by the kernel


def test_adaptive_solve_of_a_state_beyond_2_31_elements():
    """dy/dt = -y on 2^31 + 4099 elements, dopri5 from 0 to 0.5 (a func without temporaries: `torch.neg`)."""
    _need_memory(150)
    y0 = _pattern(N, 1021, 1e-3)
    nfe = [0]

    def f(t, y):
        nfe[0] += 1
        return torch.neg(y)
    with torch.no_grad():
        y = tda.odeint(f, y0, torch.tensor([0.0, 0.5], device="cuda"), rtol=1e-5, atol=1e-7)
    assert y.shape == (2, N) and nfe[0] > 8
    decay = math.exp(-0.5)
    for w in _windows(N):
        assert torch.equal(y[0][w], y0[w])
        assert torch.allclose(y[1][w], y0[w] * decay, rtol=2e-5, atol=1e-7)
