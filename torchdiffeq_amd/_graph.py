"""Captured trial steps: hipGraph capture, the ping-pong step graphs of the adaptive solvers, their reuse across solves
and the `hip_graph="auto"` safety machinery (split out of solvers.py; docs/LAB_NOTEBOOK.md §4).

  _graph_request                     the `hip_graph` option / TDEQ_HIP_GRAPH -> (wanted, auto)
  _held_tensor_ptrs, _scalar_state   what a captured graph of `func` depends on -> the cache key (_GraphStep._key)
  _side_effect_fingerprint           "auto": visible per-evaluation side effects of func (refused if any)
  _capture, _side_stream             stream capture without device synchronisation, one side stream per thread
  _GraphStep                         static buffers + two graphs (one per state side) of ONE adaptive trial step
                                     (reference semantics: rk_common.py:266-361), probe of the first replay, cache
The solvers (`solvers.py`) drive these; nothing here decides accept / reject.
"""
from __future__ import annotations

import collections
import contextlib
import functools
import gc
import os
import threading
import types
import warnings
import weakref
from typing import List, Optional

import torch

from .misc import Perturb

# hipGraph mode targets launch-latency-bound states; its stage kernel (run-time term count, scalar loads) is not the
# bandwidth-tuned one, so beyond this size the eager path is used
_GRAPH_MODE_MAX_ELEMENTS = 1 << 22


# `hip_graph="auto"`: capture only where a trial step costs launch latency rather than bandwidth (measured on the
# MI355X, profiles/r02_shard_regime.json: the captured step wins up to ~2M elements and loses beyond)
_GRAPH_AUTO_MAX_ELEMENTS = 1 << 21


# r06: the built-in default of the `hip_graph` solver option.  "auto" — captured trial steps wherever the safety net below
# lets them through (states up to _GRAPH_AUTO_MAX_ELEMENTS, a func without visible per-evaluation side effects, a replay
# that reproduces the eager step bit for bit, a func evaluation re-checked at the start of every later solve), silently
# eager everywhere else.  The reference has no such switch (torchdiffeq/_impl/odeint.py:49-108): a drop-in user never sets
# one, so the fast path has to be the one they get.  TDEQ_HIP_GRAPH=0 opts out process-wide, options={'hip_graph': False}
# per solve.
_DEFAULT_REQUEST = "auto"


def _graph_request(hip_graph):
    """(wanted, auto) from the `hip_graph` solver option: True / False, "auto", or None = the process-wide default
    taken from the environment variable TDEQ_HIP_GRAPH ("auto" — the default since r06 —, "0" or "1")."""
    if hip_graph is None:
        hip_graph = {"0": False, "": False, "1": True, "auto": "auto"}.get(
            os.environ.get("TDEQ_HIP_GRAPH", _DEFAULT_REQUEST).lower())
        if hip_graph is None:
            raise ValueError("TDEQ_HIP_GRAPH must be 0, 1 or auto")
    if isinstance(hip_graph, str):
        if hip_graph.lower() != "auto":
            raise ValueError("hip_graph must be True, False or 'auto'")
        return True, True
    return bool(hip_graph), False


_WALK_DEPTH = 4                  # containers / plain objects nested deeper than this are not searched
_WALK_ITEMS = 4096               # ... nor are the items of a container beyond this many
_PLAIN = (bool, int, float, complex, str, bytes, type(None))
# attributes every nn.Module carries for its own bookkeeping (hook dictionaries, `_parameters` ...): never user state, skipped
# by the per-solve walks (`training` is read explicitly where it matters)
_MODULE_INTERNALS = frozenset(torch.nn.Module().__dict__.keys())
_CODE_NAMES = {}                 # code object -> the global names its body (and nested code objects) mention


def _is_plain_object(value) -> bool:
    """An INSTANCE that merely stores things (a config namespace, a dataclass, a hand-written parameter holder): has a
    `__dict__`, is not a class, a Python module, a function or any other callable code object."""
    return (getattr(value, "__dict__", None) is not None and not isinstance(value, (type, types.ModuleType))
            and not hasattr(value, "__code__") and not hasattr(getattr(value, "__func__", None), "__code__")
            and not isinstance(value, (functools.partial, types.BuiltinFunctionType)))


def _request_is_explicit(hip_graph) -> bool:
    """Whether captured steps were ASKED for — by the solver option or by the environment variable — rather than being
    the built-in default: only then is a refusal worth a warning (the reference's user never heard of hipGraphs)."""
    return hip_graph is not None or "TDEQ_HIP_GRAPH" in os.environ


def _stream_is_capturing() -> bool:
    """The caller is inside a stream capture of its own (`torch.cuda.graph(...)` around the solve): the solver must not
    open a second one — its kernels simply become nodes of the caller's graph on the eager path."""
    try:
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    except Exception:
        return False


def _tensors_in(value, _level=0, _seen=None):
    """Every tensor reachable from `value` through containers (list / tuple / set / dict / deque), nn.Modules (parameters,
    buffers, plain attributes of every submodule) and plain objects' attributes — bounded in depth and width, each
    object visited once.  (r06, advisor r05: r05 stopped one container level down and never entered a Module or an
    object found in an attribute, so `self.net = nn.Linear(...)` on a plain callable object was invisible.)"""
    if isinstance(value, torch.Tensor):
        return [value]
    if isinstance(value, _PLAIN) or _level > _WALK_DEPTH:
        return []
    seen = set() if _seen is None else _seen
    if id(value) in seen:
        return []
    seen.add(id(value))
    if isinstance(value, torch.nn.Module):
        # one pass over the module tree (registered parameters and buffers in registration order, then each module's own
        # plain attributes) — `parameters()` + `buffers()` + `modules()` walk it three times with name bookkeeping, which was
        # most of a solve's cache-key cost
        out, rest, stack = [], [], [value]
        while stack:
            m = stack.pop()
            if m is not value and id(m) in seen:
                continue
            seen.add(id(m))
            out += [p for p in m._parameters.values() if p is not None]
            out += [b for b in m._buffers.values() if b is not None]
            for name, v in m.__dict__.items():
                if name not in _MODULE_INTERNALS and not isinstance(v, _PLAIN):
                    rest.append(v)
            stack += [c for c in reversed(list(m._modules.values())) if c is not None]
        for v in rest:
            out += _tensors_in(v, _level + 1, seen)
        return out
    if isinstance(value, (list, tuple, set, frozenset, collections.deque)):
        out = []
        for i, v in enumerate(value):
            if i >= _WALK_ITEMS:
                break
            out += _tensors_in(v, _level + 1, seen)
        return out
    if isinstance(value, dict):
        out = []
        for i, v in enumerate(list(value.values())):
            if i >= _WALK_ITEMS:
                break
            out += _tensors_in(v, _level + 1, seen)
        return out
    if _is_plain_object(value):
        out = []
        for v in list(value.__dict__.values()):
            out += _tensors_in(v, _level + 1, seen)
        return out
    return []


def _object_tensors(obj, _seen=None):
    """The tensors an object's attributes hold (directly, inside containers, Modules or nested plain objects)."""
    out = []
    seen = set() if _seen is None else _seen
    for v in list((getattr(obj, "__dict__", None) or {}).values()):
        out += _tensors_in(v, 1, seen)
    return out


def _named_globals(inner):
    """The module-level objects a function's body (and the lambdas / inner functions defined in it) names."""
    code, glob = getattr(inner, "__code__", None), getattr(inner, "__globals__", None)
    if code is None or glob is None:
        return []
    names, stack = [], [code]
    while stack:
        c = stack.pop()
        names += [n for n in c.co_names if n in glob]
        stack += [k for k in c.co_consts if isinstance(k, types.CodeType)]
    out, done = [], set()
    for n in names:
        if n not in done:
            done.add(n)
            out.append(glob[n])
    return out


def _held_tensors(fn, _depth=0, _seen=None):
    """The tensors a func object visibly holds: an nn.Module's parameters, buffers and plain attributes (all submodules,
    nested containers / objects / Modules inside them); a function's closure cells, defaults and the module-level
    tensors, Modules, containers, plain objects and helper functions its body names; a bound method's owner; an instance
    with `__call__` (its attributes and what its `__call__` closes over); a functools.partial's arguments."""
    seen = set() if _seen is None else _seen
    if isinstance(fn, torch.nn.Module):
        return _tensors_in(fn, 0, seen)
    if _depth > 2 or id(fn) in seen:
        return []
    seen.add(id(fn))
    held_t = []
    owner = getattr(fn, "__self__", None)
    if owner is not None and not isinstance(owner, (type, types.ModuleType)):
        held_t += _tensors_in(owner, 0, seen) if isinstance(owner, torch.nn.Module) else _object_tensors(owner, seen)
    is_function = hasattr(fn, "__code__") or hasattr(getattr(fn, "__func__", None), "__code__")
    if not is_function and not isinstance(fn, type) and hasattr(type(fn), "__call__") \
            and not isinstance(fn, functools.partial) and getattr(fn, "__dict__", None) is not None:
        # a callable INSTANCE (class with __call__): what it stores, and what its __call__ is written over
        held_t += _object_tensors(fn, seen)
        call = getattr(type(fn), "__call__", None)
        if call is not None and hasattr(call, "__code__"):
            held_t += _held_tensors(call, _depth + 1, seen)
    inner = getattr(fn, "__func__", fn)
    held = [c.cell_contents for c in (getattr(inner, "__closure__", None) or ()) if _cell_is_set(c)]
    held += list(getattr(inner, "__defaults__", None) or ())
    held += list((getattr(inner, "__kwdefaults__", None) or {}).values())
    # module-level things the body names: tensors, Modules, containers (a global list of layers), plain objects (`cfg.w`),
    # helper FUNCTIONS of the user's own (not classes, not imported library modules / functions)
    for g in _named_globals(inner):
        if isinstance(g, (torch.Tensor, torch.nn.Module, list, tuple, dict, set, collections.deque)) or _is_plain_object(g):
            held.append(g)
        elif hasattr(g, "__code__") and getattr(g, "__module__", None) == getattr(inner, "__module__", None):
            held.append(g)
    held += list(getattr(fn, "args", ())) + list((getattr(fn, "keywords", None) or {}).values())     # functools.partial
    if getattr(fn, "func", None) is not None and callable(fn.func):
        held.append(fn.func)
    for v in held:
        if isinstance(v, (torch.Tensor, torch.nn.Module)):
            held_t += _tensors_in(v, 0, seen)
        elif callable(v) and not isinstance(v, type):
            held_t += _held_tensors(v, _depth + 1, seen)
        else:
            held_t += _tensors_in(v, 0, seen)
    return held_t


def _held_tensor_ptrs(fn, _depth=0):
    """Storage addresses of `_held_tensors(fn)` — part of the captured-step cache key (see _GraphStep._key)."""
    return tuple(t.data_ptr() for t in _held_tensors(fn, _depth))


def _holds_a_tensor_that_requires_grad(fn) -> bool:
    """Whether anything func can be seen to hold is part of an autograd graph or a leaf that wants a gradient — decided
    from what func HOLDS, without evaluating it (advisor r04: a probe evaluation is visible to the user — RNG state,
    counters, one more NFE — and looks at one time only).  A func in which NOTHING can be inspected (state behind
    `__slots__`, properties, a C extension object) counts as holding one: the eager path, which records a graph, is the
    safe answer (advisor r05).  What this static look misses is caught after the first evaluation by the solvers' dynamic
    guard (`OdeFunc.grad_output_seen`)."""
    try:
        if any(t.requires_grad for t in _held_tensors(fn)):
            return True
        return not _reusable_across_solves(fn)
    except Exception:       # an exotic callable: be safe, take the path that records a graph
        return True


def _reusable_across_solves(fn) -> bool:
    """Whether a captured step of `fn` may be kept for the NEXT solve.  The cache key must change when fn is re-bound
    to new storage; for a plain function / lambda / partial / Module the discovery above sees what it holds.  An
    arbitrary callable object in which NO tensor could be found (state hidden behind properties, __slots__, nested
    objects ...) gives an empty key that cannot notice a re-binding — such a func is captured per solve, unless it
    carries a `hip_graph_token`."""
    if isinstance(fn, torch.nn.Module) or getattr(fn, "hip_graph_token", None) is not None:
        return True
    if hasattr(fn, "__code__") or hasattr(getattr(fn, "__func__", None), "__code__") or isinstance(fn, functools.partial):
        return True
    if type(fn).__module__ in ("builtins", "torch") or isinstance(fn, type(torch.tanh)):
        return True         # a builtin / torch op: holds nothing
    return len(_held_tensor_ptrs(fn)) > 0


def _cell_is_set(cell) -> bool:
    try:
        cell.cell_contents
        return True
    except ValueError:
        return False


# `hip_graph="auto"`: a first capture costs about as much as a hundred eager trial steps of a small state (≈12 ms
# against 0.2 -> 0.08 ms per step, profiles/r03_config_times.json), so a (func, layout) seen for the FIRST time runs
# eagerly and is captured only once its solve has taken this many trial steps — or at the first step of the NEXT solve
# with the same key (a training loop), whichever comes first
_AUTO_CAPTURE_AFTER_STEPS = 96
_AUTO_MIN_GRID_STEPS = 24           # fixed grids: intervals below which "auto" does not capture (one capture ≈ 1 ms there)


def _global_state(inner, versions=False, _depth=0):
    """Identity of the module-level PLAIN values a function body names — numbers, flags, strings, small containers of
    them, the plain attributes of a config-like object, and (one level) the same for the user's own helper functions it
    calls.  A captured graph bakes such values into its kernel arguments exactly like an attribute `self.scale` (r06: a
    module-level `alpha = 2.0` changed between two solves must lead to a new capture, not to a replay of the old value;
    `global NFE; NFE += 1` inside func is a per-evaluation side effect).  `versions`: also the storage address and in-place
    version counter of module-level tensors (the side-effect fingerprint; the cache key has their addresses already)."""
    if inner is None or not hasattr(inner, "__code__"):
        return ()
    out = []
    code, glob = inner.__code__, inner.__globals__
    names = _CODE_NAMES.get(code)
    if names is None:
        names, stack = [], [code]
        while stack:
            c = stack.pop()
            names += list(c.co_names)
            stack += [k for k in c.co_consts if isinstance(k, types.CodeType)]
        names = _CODE_NAMES[code] = tuple(dict.fromkeys(names))
        if len(_CODE_NAMES) > 4096:
            _CODE_NAMES.clear()
    for n in names:
        if n not in glob:
            continue
        v = glob[n]
        if isinstance(v, _PLAIN):
            out.append((n, v))
        elif isinstance(v, torch.Tensor):
            if versions:
                out.append((n, v.data_ptr(), v._version))
        elif isinstance(v, (list, tuple, collections.deque)):
            out.append((n, len(v), tuple(x for x in list(v)[:64] if isinstance(x, _PLAIN))))
        elif isinstance(v, dict):
            out.append((n, len(v), tuple(x for x in list(v.values())[:64] if isinstance(x, _PLAIN))))
        elif isinstance(v, torch.nn.Module):
            out.append((n, _scalar_state(v)))
        elif _is_plain_object(v):
            sub = []
            _visible_state(v, sub, 1)
            out.append((n, tuple(x for x in sub if versions or not (len(x) == 3 and isinstance(x[1], int) and isinstance(x[2], int)))))
        elif _depth < 1 and hasattr(v, "__code__") and getattr(v, "__module__", None) == getattr(inner, "__module__", None):
            out.append((n, _global_state(v, versions, _depth + 1)))
    return tuple(out)


def _module_global_state(module, versions=False):
    """`_global_state` of the `forward` of every user-defined class among a Module's submodules (torch.nn's own layers
    name no module-level state of the user's)."""
    out, done = [], set()
    for m in module.modules():
        cls = type(m)
        if cls in done or (cls.__module__ or "").startswith("torch."):
            continue
        done.add(cls)
        fwd = getattr(cls, "forward", None)
        if fwd is not None and hasattr(fwd, "__code__"):
            out.append(_global_state(fwd, versions))
    return tuple(out)


def _visible_state(obj, out, depth=0):
    """Cheap identity of what `obj` visibly holds — plain numbers, flags, strings, container lengths, tensor storages
    with their in-place version counters — appended to `out`."""
    for name, v in list(getattr(obj, "__dict__", {}).items()):
        if isinstance(v, torch.Tensor):
            out.append((name, v.data_ptr(), v._version))
        elif isinstance(v, (bool, int, float, complex, str, bytes, type(None))):
            out.append((name, v))
        elif isinstance(v, (list, tuple, set, frozenset, collections.deque)):
            out.append((name, len(v), tuple((t.data_ptr(), t._version) for t in v if isinstance(t, torch.Tensor)),
                        tuple(x for x in list(v)[:64] if isinstance(x, (bool, int, float)))))
        elif isinstance(v, dict) and name not in ("_parameters", "_buffers", "_modules"):
            out.append((name, len(v), tuple((t.data_ptr(), t._version) for t in v.values() if isinstance(t, torch.Tensor)),
                        tuple(x for x in list(v.values())[:64] if isinstance(x, (bool, int, float)))))
        elif isinstance(v, torch.nn.Module):
            if not isinstance(obj, torch.nn.Module) and depth < 2:     # (a Module's submodules are walked by the caller)
                out.append((name, _side_effect_fingerprint(v, None)))
        elif depth < 2 and _is_plain_object(v):
            # r06: a counter kept one object down (`self.stats.nfe += 1`)
            sub = []
            _visible_state(v, sub, depth + 1)
            out.append((name, tuple(sub)))


def _side_effect_fingerprint(fn, device):
    """What an evaluation of `fn` could change OUTSIDE its return value, as far as it can be seen from here: the
    attributes of the callable (all submodules of an nn.Module; the objects a function closes over / is bound to),
    parameter and buffer version counters, and the device's random-number offset.  Equal before and after an eager
    evaluation = no visible per-evaluation side effect — the premise of replaying `fn` from a captured hipGraph, where
    its Python body does not run at all."""
    out = []
    if isinstance(fn, torch.nn.Module):
        for m in fn.modules():
            _visible_state(m, out)
        out += [(n, t.data_ptr(), t._version) for n, t in fn.named_parameters()]
        out += [(n, t.data_ptr(), t._version) for n, t in fn.named_buffers()]
        out.append(_module_global_state(fn, versions=True))
    else:
        _visible_state(fn, out)
        owner = getattr(fn, "__self__", None)
        if owner is not None and not isinstance(owner, type):
            out.append(_side_effect_fingerprint(owner, None) if isinstance(owner, torch.nn.Module) else None)
            _visible_state(owner, out)
        inner = getattr(fn, "__func__", fn)
        for c in (getattr(inner, "__closure__", None) or ()):
            if _cell_is_set(c):
                v = c.cell_contents
                if isinstance(v, torch.nn.Module):
                    out.append(_side_effect_fingerprint(v, None))
                elif isinstance(v, torch.Tensor):
                    out.append((v.data_ptr(), v._version))
                elif isinstance(v, (bool, int, float, str, type(None))):
                    out.append(v)
                elif isinstance(v, (list, dict, set)):
                    out.append(len(v))      # e.g. `nfe = [0]` / a log list a lambda appends to
                    if isinstance(v, list):
                        out.append(tuple(x for x in v if isinstance(x, (bool, int, float))))
                elif hasattr(v, "__dict__") and not callable(v):
                    _visible_state(v, out)
        call = getattr(type(fn), "__call__", None) if not hasattr(inner, "__code__") else None
        out.append(_global_state(inner if hasattr(inner, "__code__") else call, versions=True))
    if device is not None and torch.device(device).type == "cuda":
        try:
            idx = torch.device(device).index
            gen = torch.cuda.default_generators[torch.cuda.current_device() if idx is None else idx]
            out.append(("rng", gen.initial_seed(), gen.get_offset()))
        except Exception:      # a build without generator offsets: the attribute checks still stand
            pass
    return tuple(out)


def _scalar_state(fn):
    """The plain Python values `fn` visibly holds (numbers, flags, strings, container lengths — no tensors: those are in
    the key by storage address).  A captured graph bakes such values into its kernel arguments, so they are part of the
    captured-step cache key: `self.scale = 0.5` changed between two solves leads to a new capture, not to a replay
    with the old value."""
    def plain(obj, out, depth=0):
        skip = _MODULE_INTERNALS if isinstance(obj, torch.nn.Module) else ()
        for name, v in list((getattr(obj, "__dict__", None) or {}).items()):
            if name in skip:
                continue
            if isinstance(v, (bool, int, float, complex, str, bytes, type(None))):
                out.append((name, v))
            elif isinstance(v, (list, tuple)) and len(v) <= 64 and all(isinstance(x, (bool, int, float, str)) for x in v):
                out.append((name, tuple(v)))
            elif isinstance(v, dict) and name not in ("_parameters", "_buffers", "_modules") and len(v) <= 64:
                out.append((name, tuple((k, x) for k, x in v.items() if isinstance(k, str) and isinstance(x, (bool, int, float, str)))))
            elif depth < 2 and isinstance(v, torch.nn.Module) and not isinstance(obj, torch.nn.Module):
                out.append((name, _scalar_state(v)))
            elif depth < 2 and _is_plain_object(v) and not isinstance(v, torch.nn.Module):
                sub = []
                plain(v, sub, depth + 1)        # (`self.cfg.scale`)
                out.append((name, tuple(sub)))
    out = []
    if isinstance(fn, torch.nn.Module):
        for m in fn.modules():
            plain(m, out)
            out.append(m.training)
        out.append(_module_global_state(fn))
        return tuple(out)
    plain(fn, out)
    owner = getattr(fn, "__self__", None)
    if owner is not None and not isinstance(owner, type):
        out.append(_scalar_state(owner) if isinstance(owner, torch.nn.Module) else None)
        plain(owner, out)
    inner = getattr(fn, "__func__", fn)
    out.append(_global_state(inner if hasattr(inner, "__code__") else getattr(type(fn), "__call__", None)))
    for c in (getattr(inner, "__closure__", None) or ()):
        if _cell_is_set(c):
            v = c.cell_contents
            if isinstance(v, (bool, int, float, complex, str, bytes, type(None))):
                out.append(v)
            elif isinstance(v, torch.nn.Module):
                out.append(_scalar_state(v))
            elif isinstance(v, (list, tuple)) and len(v) <= 64 and all(isinstance(x, (bool, int, float, str)) for x in v):
                out.append(tuple(v))
    return tuple(out)


def _same_words(a, b) -> bool:
    """Equality of two flat lists of host doubles, NaN == NaN."""
    return len(a) == len(b) and all(x == y or (x != x and y != y) for x, y in zip(a, b))


class _DtCell:
    """A two-double device buffer shaped like a norm plan's `ctrl_dev` ({accept, sign*dt, ...}): lets the fixed-grid
    graph mode reuse tdeq_stage_combine_dev, which reads its step size from word 1."""
    __slots__ = ("ctrl_dev",)

    def __init__(self, buf):
        self.ctrl_dev = buf


class _CaptureFailed(RuntimeError):
    """The step body could not be captured into a hipGraph; no kernel of it has run."""


_SIDE_STREAMS = threading.local()


def _side_stream(device) -> "torch.cuda.Stream":
    """The side stream of warm-up steps and stream captures: ONE per thread and device, reused.  PyTorch keeps a BLAS
    workspace (128 MiB on ROCm) per (handle, stream) that has ever run a GEMM and never returns it; a fresh
    `torch.cuda.Stream` per capture grew the process by that much for every newly captured func (measured:
    `tools/soak_training.py`, +478 MB after one captured adjoint loop), up to PyTorch's pool of 32 streams.  Per thread
    because a stream can be in one capture at a time."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    streams = getattr(_SIDE_STREAMS, "by_device", None)
    if streams is None:
        streams = _SIDE_STREAMS.by_device = {}
    stream = streams.get(key)
    if stream is None:
        stream = streams[key] = torch.cuda.Stream(torch.device("cuda", key))
    return stream


@contextlib.contextmanager
def _capture(graph, pool=None):
    """Stream capture of a step body into `graph`.  Unlike the `torch.cuda.graph` context this neither synchronises
    the device nor empties the caching allocator (both cost milliseconds — more than a short solve), and it pauses
    the cyclic garbage collector: a collection in the middle of a capture may finalize unrelated objects that own HIP
    resources (pinned buffers, events, other graphs), whose release calls are illegal while a stream is capturing."""
    current = torch.cuda.current_stream()
    side = _side_stream(current.device)
    side.wait_stream(current)
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.stream(side):
            # capture_error_mode "thread_local": only THIS thread's calls are held to the capture rules.  Under the
            # default ("global") a HIP call from any other thread during the capture is an error that invalidates it —
            # and a process with an RCCL process group has such a thread (the communicator's watchdog polls events):
            # every rank of a multi-GPU run would lose its capture at random.
            kw = {"capture_error_mode": "thread_local"}
            if pool is not None:
                kw["pool"] = pool                   # same private memory pool as a graph that never runs concurrently
            graph.capture_begin(**kw)
            try:
                yield
            finally:
                graph.capture_end()
    finally:
        if was_enabled:
            gc.enable()
    current.wait_stream(side)


class _GraphStep:
    """Static buffers + the captured hipGraphs of one adaptive trial step (RKAdaptiveStepsizeODESolver._graph_trial_step).

    r03: TWO graphs over ping-pong state buffers instead of one graph + a commit kernel.  The host reads the
    controller's decision after every replay anyway, so it — not a device-side select — knows which state pair the
    next trial step starts from:
        side 0   reads (y[0], f0),            writes y1 into y[1]; its last evaluation k0[-1] = f(t1, y[1]) IS side 1's f
        side 1   reads (y[1], k0[-1]),        writes y1 into y[0]; its last evaluation is copied into f0 (one N-word
                                              copy inside the graph: func's output buffer cannot be chosen)
    An accepted step flips the side, a rejected one replays the same graph (its input pair is untouched).  The r02
    `tdeq_step_commit` (4 reads + 4 writes per element and step, one dispatch) is gone: per accepted step 0 or 2 words
    move instead of 8.  The pair a step started from stays intact until the next replay, which is what the lazy dense
    output of the last accepted step reads.
    The first trial step runs the body eagerly on a side stream (library / allocator warm-up), the second call
    captures side 0, side 1 is captured when first needed; every later call is a replay.  Holds no reference to the
    solver (no reference cycle: the graphs and their memory pool are released by reference counting)."""

    def __init__(self, s, t0: float, dt: float):
        dev = s.y0.device
        self.y = [torch.empty_like(s.y1), torch.empty_like(s.y1)]
        self.f0 = torch.empty_like(s.y1)
        self.epart = [torch.empty_like(s.y1), torch.empty_like(s.y1)]
        self.tbuf = torch.empty(len(s._beta), dtype=s.func.time_dtype, device=dev)      # stage times: real, also for complex states
        self.ts = self.tbuf.unbind(0)
        self.k: List[Optional[List[torch.Tensor]]] = [None, None]
        self.graphs = [None, None]
        self.side = 0
        self.calls = 0
        self.plan = s.plan          # the graphs' norm kernels write into THIS plan's buffers
        self.in_use = True
        # per-element tolerances (r06): the solver's tolerance vectors are new tensors in every solve — the captured norm
        # launch reads static copies, refreshed by `reset`; 0-dim tolerances are baked in and therefore part of the key
        self.vec_tol = None
        if getattr(s, "_vec_ctrl", False):
            self.vec_tol = [v.clone() if isinstance(v, torch.Tensor) else v for v in s._vec_fused[:2]]
        self.auto = bool(getattr(s, "_graph_auto", False))    # `hip_graph="auto"`: verify before trusting replays
        self.probed = False         # a replayed step has reproduced an eager one bit for bit
        self.recheck = False        # re-used in a LATER solve (auto mode): verify the first replay, see _recheck
        self.refused = None         # why this func must not be replayed (auto mode)
        self.reset(s, t0, dt)

    # -- the current pair ----------------------------------------------------------------------------------
    def f_in(self, side: int) -> torch.Tensor:
        return self.f0 if side == 0 else self.k[0][-1]

    def reset(self, s, t0: float, dt: float) -> None:
        """Load a solve's current state into side 0's input pair: y, f(t0, y), the device-resident step state
        {accept, sign*T(dt), t0, dt} and the first trial's stage times."""
        func, kern, T = s.func, s.kernels, s.np_dtype
        self.side = 0
        self.y[0].copy_(s.y1.detach())
        self.f0.copy_(s.f1.detach())
        if self.vec_tol is not None:
            for dst, src in zip(self.vec_tol, s._vec_fused[:2]):
                if isinstance(dst, torch.Tensor):
                    dst.copy_(src)
        t0_T, dt_T, t1_T = T(t0), T(dt), T(t0 + dt)
        # (the four doubles travel in the kernel arguments of one launch: no pageable host-to-device copy per solve)
        kern.fill_scalars(self.plan.ctrl_dev, [0.0, float(dt_T) * func.sign, t0, dt])
        times = [(t1_T, Perturb.PREV) if s._alpha_is_one[i] else (t0_T + s._alpha[i] * dt_T, Perturb.NONE)
                 for i in range(len(s._beta))]
        kern.fill_scalars(self.tbuf, [func.user_time(t, p) for t, p in times])

    # -- reuse across solves ---------------------------------------------------------------------------------
    # A training loop calls odeint with the same func and state layout over and over; capturing (≈1 ms) and the eager
    # warm-up step would be paid per call.  Captured steps are therefore kept per `func` object (weakly: they go
    # away with it) and re-armed with the next solve's state.  Valid as long as func computes the same kernels on the
    # same parameter storages — what a captured graph requires anyway; `clear_graph_cache()` drops them.
    _cache = weakref.WeakKeyDictionary()
    _MAX_PER_FUNC = 4
    # r06 (captured steps are on by default): a global budget over everything kept for reuse.  A captured step owns its
    # static buffers and, through its graphs' private pool, every stage tensor of both sides — about (2 S + 9) states; a
    # process with many long-lived funcs must not grow without bound where the reference would not.  Least recently used
    # entries that no running solve holds are dropped first; TDEQ_GRAPH_CACHE_MB (default 2048) sets the budget, 0 keeps
    # nothing across solves.
    _lru = collections.OrderedDict()            # id(step) -> weakref to the cached step, oldest first
    _seen = weakref.WeakKeyDictionary()         # auto mode: func -> keys that have been solved (eagerly) once already
    _refused = weakref.WeakKeyDictionary()      # auto mode: func -> why it is never captured

    @classmethod
    def auto_policy(cls, s, seen_before: bool = False) -> str:
        """`hip_graph="auto"`, asked at the first trial step of a solve: "now" — a captured step for this (func, layout)
        is cached or the pair has been solved before (second call of a training loop): capture / replay from the first
        step; "later" — first sight: eager, captured only if this one solve turns out long
        (_AUTO_CAPTURE_AFTER_STEPS); "never" — func was found to have per-evaluation side effects.
        `seen_before`: the caller has solved this pair once already without asking (the adjoint's first backward solve
        runs with the option off, adjoint._auto_backward_due)."""
        base = s.func.base_func
        try:
            if base in cls._refused:
                return "never"
            if not _reusable_across_solves(base) or cls._budget_bytes() <= 0:
                return "later"          # (nothing can be kept for the next solve: capture only if THIS solve turns out long)
            key = s._graph_key = cls._key(s)        # (kept for `acquire`: the walk over func is not repeated within a solve)
            per_func = cls._cache.get(base)
            if per_func is not None and key in per_func:
                return "now"
            seen = cls._seen.get(base)
            if seen is None:
                seen = cls._seen[base] = set()
            if key in seen or seen_before:
                seen.add(key)
                return "now"
            seen.add(key)
        except TypeError:               # func object cannot be weakly referenced / hashed
            pass
        return "later"

    _pure = weakref.WeakSet()                   # auto mode: funcs whose first evaluation of a solve left the fingerprint alone

    @classmethod
    def status_known(cls, base) -> bool:
        """Whether `auto` has already made up its mind about this func object: refused, or seen to evaluate without a
        visible side effect."""
        try:
            return base in cls._refused or base in cls._pure
        except TypeError:
            return False

    @classmethod
    def passed_side_effect_test(cls, base) -> bool:
        try:
            return base in cls._pure and base not in cls._refused
        except TypeError:
            return False

    @classmethod
    def mark_pure(cls, base) -> None:
        try:
            cls._pure.add(base)
        except TypeError:
            pass

    @classmethod
    def refuse_func(cls, s, reason: str) -> None:
        """Auto mode, before anything was captured (r06): the FIRST evaluation of a solve — f(t0, y0), which every solve
        makes anyway — changed what the side-effect fingerprint sees.  Remembered per func object; nothing is ever
        captured for it, and the adjoint's proxy check (evaluations a counting func would see) is not run either."""
        base = s.func.base_func
        try:
            first = base not in cls._refused
            cls._refused[base] = reason
            cls._cache.pop(base, None)
        except TypeError:
            first = True
        if first and getattr(s, "_graph_explicit", True):
            warnings.warn("hip_graph='auto': {} is not captured into a hipGraph — {}; its solves run on the eager "
                          "path (pass hip_graph=True to capture it regardless)".format(type(base).__name__, reason))

    def refuse(self, s, reason: str) -> None:
        """Auto mode found `func` unfit for replay: remember it (per func object), drop the cached graphs, say so once."""
        self.refused = reason
        base = s.func.base_func
        try:
            first = base not in self._refused
            self._refused[base] = reason
            per_func = self._cache.get(base)
            if per_func is not None:
                for k in [k for k, g in per_func.items() if g is self]:
                    del per_func[k]
        except TypeError:
            first = True
        if first and getattr(s, "_graph_explicit", True):
            # (only where captured steps were asked for: under the built-in default a refusal is silent — the reference's
            #  user never heard of hipGraphs and gets the eager path, which is what the reference is)
            warnings.warn("hip_graph='auto': {} is not captured into a hipGraph — {}; its solves run on the eager "
                          "path (pass hip_graph=True to capture it regardless)".format(type(base).__name__, reason))

    def evict(self, s) -> None:
        """Drop this captured step from the per-func cache (its capture failed: nothing to replay)."""
        try:
            per_func = self._cache.get(s.func.base_func)
            if per_func is not None:
                for k in [k for k, g in per_func.items() if g is self]:
                    del per_func[k]
        except TypeError:
            pass

    @staticmethod
    def _key(s):
        c = s._ctrl
        segs = tuple((int(sg.chunk_start), int(sg.numel), float(sg.rtol), float(sg.atol))
                     for sg in getattr(s.plan, "native_segs", s.plan.segs))
        key = (type(s).__name__, str(s.y0.dtype), str(s.y0.device), int(s.layout.total), int(s.plan.chunk), segs,
               c.safety, c.ifactor, c.dfactor, c.exponent, c.min_step, c.max_step, c.time_sign, int(c.n_norm_seg))
        # A captured graph reads the STORAGES it saw: in-place updates are fine, anything re-allocated (module.to(...),
        # a re-built layer, a closure variable bound to a new tensor) must lead to a new capture — so every tensor the
        # func object can be seen to hold goes into the key; a user-supplied `hip_graph_token` attribute of func (any
        # hashable: bump it when func changes what it computes) does too.
        if getattr(s, "_vec_ctrl", False):
            # which tolerance is a vector (its VALUES are copied into the captured step's static buffers by `reset`), the
            # baked value of a 0-dim one
            key += tuple(("vec", int(v.numel())) if isinstance(v, torch.Tensor) else float(v) for v in s._vec_fused[:2])
        key += (_held_tensor_ptrs(s.func.base_func), getattr(s.func.base_func, "hip_graph_token", None),
                type(s.func).__name__, s.func.graph_key(),
                # ("auto" only: with hip_graph=True the user vouches for func, and an evaluation counter among its
                # attributes would otherwise change the key on every solve)
                _scalar_state(s.func.base_func) if getattr(s, "_graph_auto", False) else None)
        return key

    @classmethod
    def acquire(cls, s, t0: float, dt: float) -> "_GraphStep":
        if not _reusable_across_solves(s.func.base_func):
            return cls(s, t0, dt)   # nothing in the key would notice a re-bound tensor: capture per solve
        try:
            per_func = cls._cache.get(s.func.base_func)
        except TypeError:           # func object cannot be weakly referenced: no reuse
            return cls(s, t0, dt)
        key = getattr(s, "_graph_key", None) or cls._key(s)
        g = per_func.get(key) if per_func is not None else None
        if g is not None and not g.in_use:
            g.in_use = True
            cls._touch(g)
            g.auto = bool(getattr(s, "_graph_auto", False))
            g.recheck = g.auto      # "auto": the first replay of this solve is checked against one eager evaluation
            s.plan = g.plan         # read-backs must poll the buffers the captured kernels write
            g.reset(s, t0, dt)
            return g
        g = cls(s, t0, dt)
        if per_func is None:
            try:
                per_func = cls._cache[s.func.base_func] = {}
            except TypeError:
                return g
        if key not in per_func:
            if len(per_func) >= cls._MAX_PER_FUNC:       # evict the oldest entry that no running solve holds
                for old_key, old in list(per_func.items()):
                    if not old.in_use:
                        del per_func[old_key]
                        break
            if len(per_func) < cls._MAX_PER_FUNC and cls._make_room(g):
                per_func[key] = g
                cls._touch(g)
        return g

    def release(self) -> None:
        self.in_use = False

    # -- the global budget ------------------------------------------------------------------------------------
    def approx_bytes(self) -> int:
        """Device memory a cached step pins, roughly: y (2), f0, epart (2), two sides of S + 1 stage tensors and as many
        stage inputs in the graphs' pool, the tolerance copies of the per-element mode."""
        state = self.y[0].numel() * self.y[0].element_size()
        n_stage = len(self.tbuf) + 1
        extra = sum(v.numel() * v.element_size() for v in (self.vec_tol or ()) if isinstance(v, torch.Tensor))
        return state * (5 + 4 * n_stage) + extra

    @staticmethod
    def _budget_bytes() -> int:
        try:
            return int(float(os.environ.get("TDEQ_GRAPH_CACHE_MB", "2048")) * (1 << 20))
        except ValueError:
            return 2048 << 20

    @classmethod
    def _touch(cls, g) -> None:
        cls._lru.pop(id(g), None)
        cls._lru[id(g)] = weakref.ref(g)

    @classmethod
    def _cached_steps(cls):
        """(live cached steps oldest first, total bytes); entries whose step or func has gone are forgotten on the way."""
        live, total = [], 0
        for ident, ref in list(cls._lru.items()):
            g = ref()
            if g is None or not g._is_cached():
                del cls._lru[ident]
                continue
            live.append(g)
            total += g.approx_bytes()
        return live, total

    def _is_cached(self) -> bool:
        try:
            return any(g is self for per_func in self._cache.values() for g in per_func.values())
        except RuntimeError:        # the weak dictionary changed size under the iteration: count it as cached, re-examined later
            return True

    @classmethod
    def _make_room(cls, new) -> bool:
        """Evict least-recently-used cached steps that no solve holds until `new` fits the budget; False: `new` itself does
        not fit (it then serves its own solve only)."""
        budget, need = cls._budget_bytes(), new.approx_bytes()
        if need > budget:
            return False
        live, total = cls._cached_steps()
        for old in live:
            if total + need <= budget:
                break
            if old.in_use:
                continue
            for per_func in list(cls._cache.values()):
                for k in [k for k, g in per_func.items() if g is old]:
                    del per_func[k]
            cls._lru.pop(id(old), None)
            total -= old.approx_bytes()
        return total + need <= budget

    def body(self, s, side: int) -> None:
        func, kern, plan = s.func, s.kernels, s.plan
        beta, fuse, fsal = s._beta, s._fuse, s.tableau.fsal_solution
        y_cur, f_cur, y1, epart = self.y[side], self.f_in(side), self.y[1 - side], self.epart[side]
        k = [f_cur]
        yi = torch.empty_like(y_cur)
        kern.stage_combine_dev(yi, None, y_cur, [f_cur], beta[0].coef, None, plan)
        k.append(func.eval_at(self.ts[0], yi))
        n_rows = len(beta)
        carry = s._carry if hasattr(kern, "stage_combine_multi_dev") else None
        if carry is not None:
            # planned launches (tableaus.carry_plan) with the step size read on the device: same stage inputs, fewer
            # bytes and — dopri8 — one node fewer per captured step
            held, R = {}, len(carry.ops)
            for i in range(1, R):
                op = carry.ops[i]
                row = beta[i] if i < n_rows else s._c_sol
                if op is None:
                    yi = held.pop(i)
                elif len(op.targets) == 1 and not op.continues:
                    yi = y1 if i == R - 1 else torch.empty_like(y_cur)
                    kern.stage_combine_dev(yi, None, y_cur, [k[j] for j in row.idx], row.coef, None, plan)
                elif op.targets == (i, R) and i == R - 1 and not op.continues and op.idx == row.idx:
                    yi, held[R] = y1, epart
                    kern.stage_combine_dev(yi, epart, y_cur, [k[j] for j in row.idx], row.coef, fuse[0], plan)
                else:
                    # the step's solution (launch row R - 1) and the partial error go to the static buffers
                    outs = [y1 if t == R - 1 else (epart if t == R else torch.empty_like(y_cur)) for t in op.targets]
                    kern.stage_combine_multi_dev(outs, op.spec, y_cur, held.pop(i) if op.continues else None,
                                                 [k[j] for j in op.idx], plan)
                    yi = outs[0]
                    for tgt, buf in zip(op.targets[1:], outs[1:]):
                        held[tgt] = buf
                if i < n_rows:
                    k.append(func.eval_at(self.ts[i], yi))
            assert held.pop(R) is epart and not held
            if not self._norm_ctrl(s, epart, y_cur, y1, [k[j] for j in carry.err_idx], carry.err_coef,
                                   self.f0 if side == 1 and carry.err_idx and carry.err_idx[-1] == len(k) - 1 else None) \
                    and side == 1:
                self.f0.copy_(k[-1])
            self.k[side] = k
            return
        if fuse is None:
            # 16-bit states (csrc/tdeq_kernels_lp.hpp): every row whole and the error row whole — a reduced-precision row sum
            # is rounded once, so there is no partial error to hand from the last combine to the norm launch
            for i in range(1, n_rows):
                row = beta[i]
                yi = y1 if (i == n_rows - 1 and fsal) else torch.empty_like(y_cur)
                kern.stage_combine_dev(yi, None, y_cur, [k[j] for j in row.idx], row.coef, None, plan)
                k.append(func.eval_at(self.ts[i], yi))
            if not fsal:
                kern.stage_combine_dev(y1, None, y_cur, [k[j] for j in s._c_sol.idx], s._c_sol.coef, None, plan)
            err = s._c_err
            kern.error_norm_ctrl(plan, y_cur, y1, [k[j] for j in err.idx], err.coef, 0.0, s._ctrl, self.tbuf,
                                 state_in_dev=True)
            if side == 1:
                self.f0.copy_(k[-1])
            self.k[side] = k
            return
        for i in range(1, n_rows):
            row = beta[i]
            ks = [k[j] for j in row.idx]
            if i == n_rows - 1 and fsal:
                yi = y1
                kern.stage_combine_dev(yi, epart, y_cur, ks, row.coef, fuse[0], plan)
            else:
                yi = torch.empty_like(y_cur)
                kern.stage_combine_dev(yi, None, y_cur, ks, row.coef, None, plan)
            k.append(func.eval_at(self.ts[i], yi))
        if not fsal:
            sol = s._c_sol
            kern.stage_combine_dev(y1, epart, y_cur, [k[j] for j in sol.idx], sol.coef, fuse[0], plan)
        # side 0 reads its derivative from a buffer of its own (see the class text): side 1's last evaluation goes there —
        # written by the norm launch itself where that launch reads the stream anyway (r06), else by a copy node
        if not self._norm_ctrl(s, epart, y_cur, y1, [k[j] for j in fuse[1]], fuse[2],
                               self.f0 if side == 1 and fuse[1] and fuse[1][-1] == len(k) - 1 else None) and side == 1:
            self.f0.copy_(k[-1])
        self.k[side] = k

    def _norm_ctrl(self, s, epart, y_cur, y1, ks, coefs, copy_last_to=None) -> bool:
        """The step's last launch pair: the error norm continuing `epart` + the device controller, step state in device
        memory — with scalar tolerances (tdeq_error_norm_partial_ctrl) or per-element ones (tdeq_error_norm_vec_ctrl).
        `copy_last_to`: also write ks[-1] there; returns whether that was done."""
        if self.vec_tol is not None:
            s.kernels.error_norm_vec_ctrl(s.plan, y_cur, y1, ks, coefs, 0.0, self.vec_tol[0], self.vec_tol[1], s._ctrl,
                                          self.tbuf, partial=epart, state_in_dev=True)
            return False
        fused = copy_last_to is not None and getattr(s.kernels, "norm_copies_last_stage", False) \
            and not y_cur.is_complex() and os.environ.get("TDEQ_NORM_COPY", "1") != "0"
        if fused:
            s.kernels.error_norm_partial_ctrl(s.plan, epart, y_cur, y1, ks, coefs, 0.0, s._ctrl, self.tbuf,
                                              state_in_dev=True, copy_last_to=copy_last_to)
        else:
            s.kernels.error_norm_partial_ctrl(s.plan, epart, y_cur, y1, ks, coefs, 0.0, s._ctrl, self.tbuf,
                                              state_in_dev=True)
        return fused

    def run(self, s) -> None:
        """One trial step from the current side's pair.  The caller flips `side` when the step was accepted."""
        kern, func = s.kernels, s.func
        self.calls += 1
        side = self.side
        if self.calls == 1:
            before = _side_effect_fingerprint(func.base_func, s.y0.device) if self.auto else None
            self._eager_body(s, 0)
            self.eager = True
            if self.auto and _side_effect_fingerprint(func.base_func, s.y0.device) != before:
                self.refuse(s, "evaluating it changed its own attributes, buffers or the device's random-number state "
                               "(an evaluation counter, a cache, dropout ...), which a replay would not repeat")
            return
        self.eager = False
        if self.graphs[side] is None:
            graph = torch.cuda.CUDAGraph()
            nfe = func.nfe
            # the two sides never run concurrently and keep their own results alive (self.k): one memory pool serves
            # both captures — the second one reuses the blocks the first one's freed intermediates left behind instead
            # of paying for fresh device allocations
            other = self.graphs[1 - side]
            try:
                with _capture(graph, pool=None if other is None else other.pool()):
                    self.body(s, side)
            except Exception as exc:       # func is not capturable (host sync, unsupported op ...): nothing has run
                func.nfe = nfe
                raise _CaptureFailed(repr(exc)) from exc
            func.nfe = nfe
            self.graphs[side] = graph
            if self.auto and not self.probed:
                self._probe(s, side, graph)
                func.nfe += len(s._beta)
                return
        if self.recheck:
            self.recheck = False
            if not self._recheck(s, side):
                func.nfe += len(s._beta)
                return
        else:
            kern.arm_readback(s.plan)
            self.graphs[side].replay()
        func.nfe += len(s._beta)

    def _recheck(self, s, side: int) -> bool:
        """Auto mode, first replay of a LATER solve with a cached graph (r06, the price of being the default): the cache key
        holds every tensor storage and every plain value `func` can be SEEN to hold, but a Python number behind a property,
        in a C-extension object or in another module is baked into the captured kernels' arguments all the same.  So the
        step's first stage is also evaluated eagerly — one combine launch and ONE evaluation of func, with the Python body
        as it is NOW — and must equal the graph's own first evaluation bit for bit.  If not, the graph is stale: the full
        probe puts the eagerly evaluated step in place and func is not replayed again.  Returns whether the replay stands."""
        kern, func, plan = s.kernels, s.func, s.plan
        graph = self.graphs[side]
        ctrl0, times0 = plan.ctrl_dev.clone(), self.tbuf.clone()
        y_cur, f_cur = self.y[side], self.f_in(side)
        yi = torch.empty_like(y_cur)
        kern.stage_combine_dev(yi, None, y_cur, [f_cur], s._beta[0].coef, None, plan)     # (reads dt before the replay moves it on)
        kern.arm_readback(plan)
        graph.replay()
        # the eager evaluation is issued BEHIND the replay: its Python runs while the GPU replays the step (the first stage
        # time comes from the copy — the replay's controller has moved the live one on by the time these kernels run)
        nfe = func.nfe
        k1 = func.eval_at(times0[0], yi)
        func.nfe = nfe
        if torch.equal(k1, self.k[side][1]):
            return True
        plan.ctrl_dev.copy_(ctrl0)
        self.tbuf.copy_(times0)
        self.probed = False
        self._probe(s, side, graph)
        if self.refused is None:
            # (the full probe agrees although the single evaluation did not: a func that is not deterministic from call to
            #  call — nothing a replay could be trusted with either)
            self.refuse(s, "two evaluations of it at the same (t, y) differ")
        return False

    def _eager_body(self, s, side: int) -> None:
        current = torch.cuda.current_stream(s.y0.device)
        stream = _side_stream(s.y0.device)
        stream.wait_stream(current)
        with torch.cuda.stream(stream):
            self.body(s, side)
        current.wait_stream(stream)

    def _probe(self, s, side: int, graph) -> None:
        """Auto mode, once per captured func: the trial step at hand is run TWICE from the same state — evaluated
        eagerly, then replayed from the fresh graph — and everything a step produces must agree bit for bit: y1, the
        partial error, the controller's decision words on the host and on the device, the next stage times.  On
        agreement the replay's results stay in place (they live in the buffers the graphs are wired to).  Otherwise the
        eager results are put back (`take_words` hands the caller their decision words) — a mismatch costs nothing but
        the capture: the solve goes on eagerly and `func` is not captured again."""
        kern, func, plan = s.kernels, s.func, s.plan
        k_graph = self.k[side]                      # the stage tensors the captured nodes write (the graph's own pool)
        ctrl0, times0 = plan.ctrl_dev.clone(), self.tbuf.clone()
        nfe = func.nfe
        self._eager_body(s, side)
        func.nfe = nfe
        accept, dt_next, ratio, bad = kern.read_ctrl(plan)
        words_e = (accept, dt_next, ratio, list(bad))
        k_eager = self.k[side]
        y1_e, ep_e = self.y[1 - side].clone(), self.epart[side].clone()
        ctrl_e, times_e = plan.ctrl_dev.clone(), self.tbuf.clone()
        plan.ctrl_dev.copy_(ctrl0)
        self.tbuf.copy_(times0)
        self.k[side] = k_graph
        kern.arm_readback(plan)
        graph.replay()
        accept, dt_next, ratio, bad = kern.read_ctrl(plan)
        flat = lambda w: [float(w[0]), w[1], w[2]] + list(w[3])
        same = (_same_words(flat((accept, dt_next, ratio, bad)), flat(words_e)) and torch.equal(y1_e, self.y[1 - side])
                and torch.equal(ep_e, self.epart[side]) and torch.equal(times_e, self.tbuf)
                and _same_words(ctrl_e.tolist(), plan.ctrl_dev.tolist())
                and all(torch.equal(a, b) for a, b in zip(k_eager[1:], k_graph[1:])))
        self.probed = True
        if not same:
            self.y[1 - side].copy_(y1_e)
            self.epart[side].copy_(ep_e)
            plan.ctrl_dev.copy_(ctrl_e)
            self.tbuf.copy_(times_e)
            self.k[side] = k_eager
            self._words = words_e
            self.refuse(s, "a replayed trial step did not reproduce the eagerly evaluated one bit for bit (func is not a "
                           "pure function of t, y and its parameters)")

    def take_words(self, kern, plan):
        """(accept, dt_next, error_ratio, nonfinite) of the step `run` just took."""
        words, self._words = getattr(self, "_words", None), None
        return words if words is not None else kern.read_ctrl(plan)

    def accepted(self, s) -> None:
        """The step just run was accepted: its end state becomes the next trial step's input pair."""
        if self.eager:
            # the warm-up step ran on transient buffers: move its end state into side 0's pair (once per capture)
            self.y[0].copy_(self.y[1])
            self.f0.copy_(self.k[0][-1])
            return
        self.side = 1 - self.side


def clear_graph_cache() -> None:
    """Drop every captured trial-step graph kept for reuse (options={'hip_graph': True})."""
    _GraphStep._cache.clear()
    _GraphStep._lru.clear()
