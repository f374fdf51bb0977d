"""The host-side safety net behind `hip_graph='auto'` as the built-in default (r06), as far as it can run without a GPU:
what the static look at func finds (advisor r05), what goes into the cache key and the side-effect fingerprint, the default
request, and the global budget over cached captured steps.  The captured paths themselves: tests/test_graph_default_gpu.py."""
import functools
import types

import pytest
import torch

from torchdiffeq_amd import _graph

_CFG = types.SimpleNamespace(w=torch.randn(2, 2, requires_grad=True), scale=1.0)
_LAYERS = [torch.nn.Linear(2, 2)]
_ALPHA = 1.0
_COUNT = 0


class _Holder:
    def __init__(self):
        self.net = torch.nn.Linear(2, 2)
        self.stats = types.SimpleNamespace(n=0)
        self.cfg = types.SimpleNamespace(scale=1.0)

    def __call__(self, t, y):
        self.stats.n += 1
        return self.net(y) * self.cfg.scale

    def rhs(self, t, y):
        return self.net(y)


def _reads_config(t, y):
    return y @ _CFG.w * _CFG.scale


def _indexes_layers(t, y):
    return _LAYERS[0](y)


def _helper(y):
    return y @ _CFG.w


def _calls_helper(t, y):
    return _helper(y)


def _reads_alpha(t, y):
    return -_ALPHA * y


def _counts_globally(t, y):
    global _COUNT
    _COUNT += 1
    return -y


def test_default_request_is_auto(monkeypatch):
    monkeypatch.delenv("TDEQ_HIP_GRAPH", raising=False)
    assert _graph._graph_request(None) == (True, True) and not _graph._request_is_explicit(None)
    assert _graph._graph_request(False) == (False, False) and _graph._request_is_explicit(False)
    monkeypatch.setenv("TDEQ_HIP_GRAPH", "0")
    assert _graph._graph_request(None) == (False, False) and _graph._request_is_explicit(None)
    monkeypatch.setenv("TDEQ_HIP_GRAPH", "sometimes")
    with pytest.raises(ValueError):
        _graph._graph_request(None)


def test_static_look_finds_parameters_the_r05_walk_missed():
    """Advisor r05: a plain callable object holding a Module, its bound method, a function reading a global config
    object, one indexing a global list of Modules, one calling a helper that does — and what cannot be inspected at all
    counts as holding a parameter."""
    h = _Holder()
    for f in (h, h.rhs, _reads_config, _indexes_layers, _calls_helper, functools.partial(_reads_config)):
        assert _graph._holds_a_tensor_that_requires_grad(f), f
    assert not _graph._holds_a_tensor_that_requires_grad(lambda t, y: -y)
    assert not _graph._holds_a_tensor_that_requires_grad(torch.tanh)
    w = h.net.weight

    class Opaque:
        __slots__ = ("_w",)

        def __init__(self):
            self._w = w

        def __call__(self, t, y):
            return y @ self._w
    assert _graph._holds_a_tensor_that_requires_grad(Opaque())
    # the key sees the storages behind all of these
    assert w.data_ptr() in _graph._held_tensor_ptrs(h) and w.data_ptr() in _graph._held_tensor_ptrs(h.rhs)
    assert _CFG.w.data_ptr() in _graph._held_tensor_ptrs(_reads_config)
    assert _LAYERS[0].weight.data_ptr() in _graph._held_tensor_ptrs(_indexes_layers)
    assert _CFG.w.data_ptr() in _graph._held_tensor_ptrs(_calls_helper)


def test_plain_values_behind_one_indirection_are_in_the_key_and_the_fingerprint():
    global _ALPHA
    k = _graph._scalar_state(_reads_alpha)
    _ALPHA = 2.0
    try:
        assert _graph._scalar_state(_reads_alpha) != k                  # a module-level number
    finally:
        _ALPHA = 1.0
    k = _graph._scalar_state(_reads_config)
    _CFG.scale = 3.0
    try:
        assert _graph._scalar_state(_reads_config) != k                 # an attribute of a module-level config object
    finally:
        _CFG.scale = 1.0
    h = _Holder()
    k = _graph._scalar_state(h)
    h.cfg.scale = 0.5
    assert _graph._scalar_state(h) != k                                 # ... of a nested config object

    class M(torch.nn.Module):
        def forward(self, t, y):
            return -_ALPHA * y
    m = M()
    k = _graph._scalar_state(m)
    _ALPHA = 4.0
    try:
        assert _graph._scalar_state(m) != k                             # a Module's forward naming a module-level number
    finally:
        _ALPHA = 1.0
    # per-evaluation side effects one object down / at module level
    before = _graph._side_effect_fingerprint(h, None)
    h(torch.tensor(0.0), torch.ones(2))
    assert _graph._side_effect_fingerprint(h, None) != before
    before = _graph._side_effect_fingerprint(_counts_globally, None)
    _counts_globally(0.0, torch.ones(1))
    assert _graph._side_effect_fingerprint(_counts_globally, None) != before
    pure = torch.nn.Linear(2, 2)
    before = _graph._side_effect_fingerprint(pure, None)
    pure(torch.ones(2))
    assert _graph._side_effect_fingerprint(pure, None) == before


def test_module_bookkeeping_attributes_are_not_walked_per_solve():
    assert {"_parameters", "_buffers", "_modules", "training"} <= set(_graph._MODULE_INTERNALS)
    m = torch.nn.Sequential(torch.nn.Linear(2, 2), torch.nn.Tanh())
    state = _graph._scalar_state(m)
    assert all(not (isinstance(e, tuple) and e and e[0] in _graph._MODULE_INTERNALS) for e in state)
    m.train(False)
    assert _graph._scalar_state(m) != state                             # `training` itself is part of the key


class _FakeStep:
    def __init__(self, nbytes):
        self.nbytes, self.in_use = nbytes, False

    def approx_bytes(self):
        return self.nbytes

    def _is_cached(self):
        return any(g is self for per_func in _graph._GraphStep._cache.values() for g in per_func.values())


def test_cached_steps_are_evicted_least_recently_used_first_within_the_budget(monkeypatch):
    G = _graph._GraphStep
    _graph.clear_graph_cache()
    monkeypatch.setenv("TDEQ_GRAPH_CACHE_MB", str(250 / (1 << 20) * 1.0))           # a budget of 250 bytes
    funcs = [torch.nn.Identity() for _ in range(4)]
    steps = [_FakeStep(100) for _ in range(4)]
    try:
        for f, g in zip(funcs[:2], steps[:2]):
            assert G._make_room(g)
            G._cache[f] = {"k": g}
            G._touch(g)
        live, total = G._cached_steps()
        assert live == steps[:2] and total == 200
        G._touch(steps[0])                                  # step 0 is used again: step 1 is now the oldest
        assert G._make_room(steps[2])
        G._cache[funcs[2]] = {"k": steps[2]}
        G._touch(steps[2])
        live, total = G._cached_steps()
        assert live == [steps[0], steps[2]] and total == 200 and not G._cache[funcs[1]]
        steps[0].in_use = True                              # a running solve holds step 0: only step 2 can go
        assert G._make_room(steps[3])
        assert G._cache[funcs[0]] and not G._cache[funcs[2]]
        assert not G._make_room(_FakeStep(300))             # larger than the whole budget: never cached
        monkeypatch.setenv("TDEQ_GRAPH_CACHE_MB", "0")
        assert not G._make_room(_FakeStep(1))
    finally:
        _graph.clear_graph_cache()
