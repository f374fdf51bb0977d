"""Odd-but-legal inputs on the HIP path vs the package's own host path on the CPU — which reproduces the reference bit for
bit (tests/test_hostpath.py, tools/api_diff_vs_reference_inputs.py: the same cases against the imported reference, all
SAME).  Non-contiguous / expanded / non-leaf initial states, tuple states of mixed dtypes, fp64 time grids over fp32
states, decreasing grids, per-component tolerances, output counts in the hundreds.  fp32: 2e-5 relative to the largest
value (the user's GEMM rounds differently on the two devices), fp64: 1e-10."""
import warnings

import pytest
import torch

import torchdiffeq_amd as tda

pytestmark = pytest.mark.gpu


class _Field(torch.nn.Module):
    def __init__(self, dtype=torch.float32):
        super().__init__()
        torch.manual_seed(0)
        self.lin = torch.nn.Linear(3, 3).to(dtype)

    def forward(self, t, y):
        return torch.tanh(self.lin(y)) * torch.cos(t)


def _gen(*shape, seed=2):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


T4 = torch.linspace(0, 1, 4)
CASES = {
    "transposed": dict(y0=lambda: _gen(3, 6).T),
    "strided": dict(y0=lambda: _gen(12, 3)[::2]),
    "expanded": dict(y0=lambda: _gen(1, 3).expand(6, 3)),
    "nonleaf_grad": dict(y0=lambda: _gen(6, 3).requires_grad_(True) * 2),
    "nonleaf_grad_adjoint": dict(y0=lambda: _gen(6, 3).requires_grad_(True) * 2, adjoint=True),
    "301_outputs": dict(y0=lambda: _gen(6, 3), t=torch.linspace(0, 1, 301)),
    "decreasing": dict(y0=lambda: _gen(6, 3), t=torch.linspace(1, 0, 7)),
    "decreasing_adjoint": dict(y0=lambda: _gen(6, 3), t=torch.linspace(1, -1, 7), adjoint=True),
    "t64_y32_dopri5": dict(y0=lambda: _gen(6, 3), t=T4.double()),
    "t64_y32_rk4": dict(y0=lambda: _gen(6, 3), t=T4.double(), method="rk4"),
    "t64_y32_adjoint": dict(y0=lambda: _gen(6, 3), t=T4.double(), adjoint=True),
    "t32_y64": dict(y0=lambda: _gen(6, 3).double(), dtype=torch.float64),
    "t32_y64_adjoint": dict(y0=lambda: _gen(6, 3).double(), dtype=torch.float64, adjoint=True),
    "dtype_option": dict(y0=lambda: _gen(6, 3), t=T4.double(), options=dict(dtype=torch.float32)),
    "tsit5_decreasing": dict(y0=lambda: _gen(6, 3), t=torch.linspace(1, 0, 4), method="tsit5"),
    "dopri8": dict(y0=lambda: _gen(6, 3), method="dopri8"),
    "first_step_past_the_end": dict(y0=lambda: _gen(6, 3), options=dict(first_step=5.0)),
}


def _solve(case, device):
    kw = dict(case)
    dtype = kw.pop("dtype", torch.float32)
    f = _Field(dtype).to(device)
    x = kw.pop("y0")().to(device)
    t = kw.pop("t", T4).to(device)
    fn = tda.odeint_adjoint if kw.pop("adjoint", False) else tda.odeint
    with warnings.catch_warnings():
        if device == "cpu":
            warnings.simplefilter("ignore", tda.HostPathWarning)
        else:
            warnings.simplefilter("error", tda.HostPathWarning)
        out = fn(f, x, t, method=kw.pop("method", "dopri5"), **kw)
        res = [out.detach().cpu()]
        if out.requires_grad:
            out[-1].pow(2).sum().backward()
            res += [p.grad.cpu() for p in f.parameters()]
    return res


@pytest.mark.parametrize("name", sorted(CASES))
def test_odd_input_on_the_hip_path_matches_the_host_path(name):
    got, want = _solve(CASES[name], "cuda"), _solve(CASES[name], "cpu")
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g.shape == w.shape and g.dtype == w.dtype
        tol = 2e-5 if g.dtype == torch.float32 else 1e-10
        assert float((g.double() - w.double()).abs().max()) <= tol * float(w.double().abs().max()), name


def test_tuple_state_of_mixed_dtypes_and_per_component_tolerances():
    def f(t, y):
        return (-y[0], 0.3 * y[1])
    out = {}
    for device in ("cuda", "cpu"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore" if device == "cpu" else "error", tda.HostPathWarning)
            y0 = (_gen(6, 3).to(device), _gen(2, 3).double().to(device))
            out[device] = tda.odeint(f, y0, T4.to(device), rtol=(1e-3, 1e-6), atol=(1e-4, 1e-8))
    for g, w in zip(out["cuda"], out["cpu"]):
        assert g.dtype == w.dtype == torch.float64            # promoted as a whole (misc.py:206-207)
        assert torch.allclose(g.cpu(), w, rtol=0, atol=1e-12)
