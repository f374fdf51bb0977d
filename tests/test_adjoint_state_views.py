"""Unit / property tests (CPU) of the views the adjoint hands to norms and callbacks (adjoint._reference_state /
_components, AdjointBuiltinNorm) and of misc.plugin_solver_inputs — the pieces behind the drop-in fixes of r03."""
import pytest
import torch
from hypothesis import given, settings, strategies as st

from torchdiffeq_amd import adjoint as adj
from torchdiffeq_amd.misc import BuiltinNorm, StateLayout, mixed_norm, plugin_solver_inputs

shapes_st = st.lists(st.lists(st.integers(0, 4), min_size=0, max_size=3).map(tuple), min_size=1, max_size=4)


@settings(max_examples=60, deadline=None)
@given(shapes=shapes_st, n_params=st.integers(0, 3), seed=st.integers(0, 10 ** 6))
def test_reference_state_roundtrip(shapes, n_params, seed):
    g = torch.Generator().manual_seed(seed)
    fwd = StateLayout([torch.Size(s) for s in shapes], True)
    p_shapes = [torch.Size((i + 2,)) for i in range(n_params)]
    aug_l = StateLayout([torch.Size(())] + fwd.shapes + fwd.shapes + p_shapes, True, chunk=fwd.chunk)
    parts = [torch.randn(s, generator=g, dtype=torch.float64) for s in aug_l.shapes]
    flat = aug_l.pack(parts)
    t, y, a, *ps = adj._reference_state(aug_l, fwd, flat)
    n = sum(fwd.numels)
    assert t.shape == () and y.shape == (n,) and a.shape == (n,) and [p.shape for p in ps] == p_shapes
    for got, want in zip(adj._components(y, fwd), parts[1:1 + len(shapes)]):
        assert got.shape == want.shape and torch.equal(got, want)
    for got, want in zip(adj._components(a, fwd), parts[1 + len(shapes):1 + 2 * len(shapes)]):
        assert torch.equal(got, want)
    # padding of the flat buffer is zero (StateLayout.pack), never garbage
    mask = torch.ones(aug_l.total, dtype=torch.bool)
    for off, m in zip(aug_l.offsets, aug_l.numels):
        mask[off:off + m] = False
    assert bool((flat[mask] == 0).all())


def test_tensor_state_keeps_its_shape():
    fwd = StateLayout([torch.Size((2, 3))], False)
    aug_l = StateLayout([torch.Size(()), torch.Size((2, 3)), torch.Size((2, 3)), torch.Size((4,))], True, chunk=fwd.chunk)
    flat = aug_l.pack([torch.tensor(1.0), torch.ones(2, 3), 2 * torch.ones(2, 3), 3 * torch.ones(4)])
    t, y, a, p = adj._reference_state(aug_l, fwd, flat)
    assert y.shape == (2, 3) and a.shape == (2, 3) and p.shape == (4,)
    assert adj._components(y, fwd)[0] is y


@pytest.mark.parametrize("seminorm", [False, True])
def test_adjoint_builtin_norm_is_the_reference_formula(seminorm):
    fwd = StateLayout([torch.Size(()), torch.Size((2, 2))], True)
    norm = adj.AdjointBuiltinNorm(fwd, 2, seminorm)
    assert isinstance(norm, BuiltinNorm) and norm.n_skip_tail == (2 if seminorm else 0)
    t = torch.tensor(-0.3)
    y = torch.tensor([4.0, 1.0, 1.0, 1.0, 1.0])
    a = torch.tensor([0.1, 3.0, 3.0, 3.0, 3.0])
    p1, p2 = torch.full((7,), 5.0), torch.tensor(-6.0)
    got = norm((t, y, a, p1, p2))
    want = max(0.3, 4.0, 1.0, 0.1, 3.0) if seminorm else 6.0
    assert float(got) == pytest.approx(want)
    assert float(norm(torch.tensor([3.0, 4.0]))) == pytest.approx((12.5) ** 0.5)       # a bare tensor: plain RMS


def test_plugin_solver_inputs():
    class Native:
        flat_state_native = True

    class Foreign:
        pass
    lay = StateLayout([torch.Size((3,)), torch.Size((2, 2))], True)
    opts = {"norm": mixed_norm, "first_step": 0.1}
    o, r, a = plugin_solver_inputs(Native, lay, opts, (1e-3, 1e-4), 1e-6, torch.device("cpu"))
    assert o is opts and r == (1e-3, 1e-4) and a == 1e-6                      # the package's own classes: untouched
    o, r, a = plugin_solver_inputs(Foreign, lay, opts, (1e-3, 1e-4), 1e-6, torch.device("cpu"))
    assert o is not opts and o["first_step"] == 0.1 and callable(o["norm"]) and not isinstance(o["norm"], BuiltinNorm)
    assert a == 1e-6 and torch.is_tensor(r) and r.shape == (lay.total,)
    assert torch.allclose(r[:3], torch.full((3,), float(torch.as_tensor(1e-3)), dtype=torch.float64))
    off = lay.offsets[1]
    assert torch.allclose(r[off:off + 4], torch.full((4,), float(torch.as_tensor(1e-4)), dtype=torch.float64))
    flat = lay.pack([torch.tensor([3.0, 4.0, 0.0]), torch.full((2, 2), 2.0)])
    assert float(o["norm"](flat)) == pytest.approx((25.0 / 3) ** 0.5)          # max of the component RMS, padding ignored
    # a tensor state needs no adaptation at all
    lay1 = StateLayout([torch.Size((5,))], False)
    assert plugin_solver_inputs(Foreign, lay1, opts, 1e-3, 1e-6, torch.device("cpu")) == (opts, 1e-3, 1e-6)
