"""Adjoint of a model with MORE parameter tensors than the kernels' inline segment table holds (16): the augmented
state [vjp_t | y | adj_y | 24 parameter adjoints] has 27 segments — the norm kernels take the segment table from device
memory, the controller runs on the host, and `pack_fused` assembles the dynamics' outputs in ceil(27/16) = 2 launches.
Gradients must equal the un-fused packing path bit for bit and the gradients of backprop through the solver to the
solve tolerance."""
import pytest
import torch

import torchdiffeq_amd as tda


class Deep(torch.nn.Module):
    def __init__(self, dim=6, width=10, layers=12):
        super().__init__()
        torch.manual_seed(3)
        sizes = [dim] + [width] * (layers - 1) + [dim]
        # initialised on the CPU whatever the default device is: the same weights in the cpu and cuda variants
        self.layers = torch.nn.ModuleList(torch.nn.Linear(a, b, device="cpu") for a, b in zip(sizes[:-1], sizes[1:]))

    def forward(self, t, y):
        for lin in self.layers[:-1]:
            y = torch.tanh(lin(y))
        return self.layers[-1](y) * 0.5


def _grads(f, y0, t, fn, **kw):
    for p in f.parameters():
        p.grad = None
    x = y0.clone().requires_grad_(True)
    fn(f, x, t, **kw)[-1].pow(2).sum().backward()
    return [x.grad.clone()] + [p.grad.clone() for p in f.parameters()]


def test_adjoint_with_27_segments(dev, monkeypatch):
    f = Deep().double().to(dev)
    assert len(list(f.parameters())) == 24
    y0 = torch.randn(9, 6, dtype=torch.float64, device="cpu", generator=torch.Generator(device="cpu").manual_seed(1)).to(dev)
    t = torch.tensor([0.0, 0.7, 1.5], dtype=torch.float64, device=dev)
    kw = dict(rtol=1e-8, atol=1e-10, method="dopri5")
    fused = _grads(f, y0, t, tda.odeint_adjoint, **kw)
    monkeypatch.setenv("TDEQ_PACK_FUSED", "0")
    plain = _grads(f, y0, t, tda.odeint_adjoint, **kw)
    for a, b in zip(fused, plain):
        assert torch.equal(a, b)
    through = _grads(f, y0, t, tda.odeint, **kw)            # backprop through the solver
    for a, b in zip(fused, through):
        # two different discretisations of the same gradient: they agree to the solve's accuracy (measured 1.3e-5 at
        # rtol 1e-8, 3e-7 at rtol 1e-10)
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-12


@pytest.mark.gpu
def test_many_segments_lookahead_and_captured_backward(monkeypatch):
    """With 27 segments the device-resident controller (look-ahead) and the captured backward solve are available
    too (r02): same gradients as the host-driven loop (to the few-ulp difference of the two `pow`s in dt_next) and,
    captured vs eager, bit for bit."""
    f = Deep().double().cuda()
    y0 = torch.randn(9, 6, dtype=torch.float64, device="cpu", generator=torch.Generator(device="cpu").manual_seed(1)).cuda()
    t = torch.tensor([0.0, 0.7, 1.5], dtype=torch.float64, device="cuda")
    kw = dict(rtol=1e-8, atol=1e-10, method="dopri5")
    look = _grads(f, y0, t, tda.odeint_adjoint, **kw)
    tda.clear_graph_cache()
    captured = _grads(f, y0, t, tda.odeint_adjoint, options=dict(hip_graph=True), **kw)
    captured2 = _grads(f, y0, t, tda.odeint_adjoint, options=dict(hip_graph=True), **kw)
    for a, b, c in zip(captured, captured2, look):
        assert torch.equal(a, c) and torch.equal(b, c)
    from torchdiffeq_amd.solvers import _GraphStep
    assert len(_GraphStep._cache.get(f)) == 2
    tda.clear_graph_cache()
    monkeypatch.setenv("TDEQ_LOOKAHEAD", "0")
    host = _grads(f, y0, t, tda.odeint_adjoint, **kw)
    for a, b in zip(look, host):
        assert float((a - b).abs().max()) <= 1e-12 * float(b.abs().max()) + 1e-300
