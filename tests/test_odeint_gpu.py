"""End-to-end parity of odeint on the MI355X: closed forms and the CPU oracle solver."""
import numpy as np
import pytest
import scipy.linalg
import torch

import torchdiffeq_amd as tda

pytestmark = pytest.mark.gpu


def _linear_problem(B, D, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    G = torch.randn(D, D, generator=g, dtype=torch.float64) / D ** 0.5
    A = 0.5 * (G - G.T) - 0.1 * torch.eye(D, dtype=torch.float64)
    y0 = torch.randn(B, D, generator=g, dtype=torch.float64)
    return A.to(dtype), y0.to(dtype)


@pytest.mark.parametrize("method,dtype,rtol,atol,tol", [
    ("dopri5", torch.float32, 1e-7, 1e-9, 1e-5),
    ("dopri5", torch.float64, 1e-9, 1e-11, 1e-7),
    ("dopri8", torch.float64, 1e-9, 1e-11, 1e-6),   # reference: 4.0e-7 from expm at cfg4
    ("dopri8", torch.float32, 1e-6, 1e-8, 1e-5),
])
def test_linear_closed_form(method, dtype, rtol, atol, tol):
    """cfg2/cfg4-shaped linear field at reduced batch: y(t) = y0 @ expm(A t)^T."""
    A, y0 = _linear_problem(512, 128, dtype)
    t = torch.tensor([0.0, 0.4, 1.0], dtype=torch.float64)
    Ad = A.cuda()
    with torch.no_grad():
        y = tda.odeint(lambda t_, y_: y_ @ Ad.T, y0.cuda(), t.cuda(), method=method, rtol=rtol, atol=atol)
    assert y.shape == (3, 512, 128) and y.dtype == dtype
    assert torch.equal(y[0].cpu(), y0)
    for i, ti in enumerate(t.tolist()):
        exact = y0.double() @ torch.from_numpy(scipy.linalg.expm(A.double().numpy() * ti)).T
        rel = float((y[i].cpu().double() - exact).abs().max() / exact.abs().max())
        assert rel < tol, (method, dtype, ti, rel)


def test_reverse_time_round_trip():
    """Integrate 0 -> 1 and back 1 -> 0: returns to y0 (size-independent property)."""
    A, y0 = _linear_problem(2048, 64, torch.float64)
    Ad = A.cuda()
    f = lambda t_, y_: y_ @ Ad.T
    with torch.no_grad():
        y1 = tda.odeint(f, y0.cuda(), torch.tensor([0.0, 1.0]).cuda(), rtol=1e-10, atol=1e-12)[-1]
        yb = tda.odeint(f, y1, torch.tensor([1.0, 0.0]).cuda(), rtol=1e-10, atol=1e-12)[-1]
    assert float((yb.cpu() - y0).abs().max()) < 1e-8


def test_rk4_spiral_cfg1_known_answer():
    """cfg1 (examples/ode_demo.py spiral, rk4 on the t grid): reference CPU result, SURVEY.md §8c."""
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]]).cuda()
    y0 = torch.tensor([[2.0, 0.0]]).cuda()
    t = torch.linspace(0.0, 25.0, 1000).cuda()
    with torch.no_grad():
        y = tda.odeint(lambda t_, y_: (y_ ** 3) @ A, y0, t, method="rk4")
    ref = torch.tensor([-0.4436032772064209, 0.27951884269714355])
    assert torch.allclose(y[-1, 0].cpu(), ref, rtol=2e-5, atol=1e-6)


def test_tuple_state_and_mixed_norm():
    A, y0 = _linear_problem(300, 16, torch.float32)
    Ad = A.cuda()
    ya, yb = y0[:100].cuda(), y0[100:].cuda()
    with torch.no_grad():
        out = tda.odeint(lambda t_, y_: (y_[0] @ Ad.T, y_[1] @ Ad.T), (ya, yb), torch.tensor([0.0, 1.0]).cuda(),
                         rtol=1e-6, atol=1e-8)
    assert isinstance(out, tuple) and out[0].shape == (2, 100, 16) and out[1].shape == (2, 200, 16)
    exact = y0.double() @ torch.from_numpy(scipy.linalg.expm(A.double().numpy())).T
    got = torch.cat([out[0][-1], out[1][-1]]).cpu().double()
    assert float((got - exact).abs().max() / exact.abs().max()) < 1e-5


def test_cpu_state_takes_the_host_path_and_gpu_state_the_kernels():
    """r03: a CPU state is integrated by the torch-op host path (with a HostPathWarning, tests/test_hostpath.py); the
    same call with the state on the GPU runs the HIP kernels and gives the same answer to rounding."""
    import warnings
    from torchdiffeq_amd import _fallback
    y0, t = torch.linspace(-1.0, 1.0, 12).reshape(4, 3), torch.tensor([0.0, 1.0])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", _fallback.HostPathWarning)
        y_cpu = tda.odeint(lambda t_, y_: -y_, y0, t)
    y_gpu = tda.odeint(lambda t_, y_: -y_, y0.cuda(), t.cuda())
    assert y_cpu.device.type == "cpu" and y_gpu.is_cuda
    # fp32 at the default rtol 1e-7 sits on the rounding floor: the host path sums a tableau row in ATen's order (= the
    # reference, bit for bit), the kernels left to right — the two solves differ like the reference differs from itself
    # when its thread count changes (3.9e-6)
    assert torch.allclose(y_cpu, y_gpu.cpu(), rtol=2e-5, atol=1e-6)


def test_solves_on_a_user_stream_match_the_default_stream():
    """All launches go to torch's CURRENT stream: a solve issued inside `torch.cuda.stream(s)` (look-ahead path,
    adjoint included) must give the default-stream result bit for bit."""
    import torch
    import torchdiffeq_amd as tda
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 8).double().cuda()

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = lin

        def forward(self, t, y):
            return torch.tanh(self.lin(y)) * torch.cos(t)

    y0 = torch.randn(64, 8, dtype=torch.float64, device="cuda")
    t = torch.tensor([0.0, 0.8, 2.0], dtype=torch.float64, device="cuda")

    def run():
        f = F()
        f.zero_grad()
        x = y0.clone().requires_grad_(True)
        y = tda.odeint_adjoint(f, x, t, rtol=1e-7, atol=1e-9, method="dopri5")
        y[-1].pow(2).sum().backward()
        with torch.no_grad():
            plain = tda.odeint(f, y0, t, rtol=1e-7, atol=1e-9, method="dopri8")
        return y.detach().clone(), x.grad.clone(), lin.weight.grad.clone(), plain

    ref = run()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
