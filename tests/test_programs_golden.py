"""Forty training-loop PROGRAMS run by the reference (tests/golden/programs.npz <- make_golden.py programs; the programs are
those of tools/fuzz_programs_vs_reference.py, seed 7): an `nn.Module` field — time-concatenated MLP, concat-squash layer, a
CNF field differentiating inside `forward`, a linear field —, one to three iterations of solve -> loss -> backward -> SGD
step through `odeint_adjoint` or `odeint`, sometimes an event solve at the end.  The package's CPU host path replays them:
through `odeint_adjoint` EVERY logged value — solutions, losses, all gradients, evaluation counts after each iteration, the
event — must equal the reference's in every bit (each iteration starts from parameters updated with the previous
iteration's gradients, so one differing bit anywhere shows up downstream); backprop through plain `odeint` gives the
solution, loss and evaluation count bit for bit and the gradients to rounding (the package's hand-written backward adds the
cotangents in its own order, CHANGELOG.md (9))."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

from _cases import load

os.environ.setdefault("TDEQ_FUZZ_DEVICE", "cpu")        # import the program generator without the reference (GPU box / CI)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
_argv, sys.argv = sys.argv, ["fuzz_programs_vs_reference.py", "7", "40"]
try:
    import fuzz_programs_vs_reference as fz
finally:
    sys.argv = _argv

Z = load("programs.npz")
N = 40
_CASES = []


def _case(i):
    """The i-th program of seed 7 (the generator is sequential: build them all once, in order)."""
    if not _CASES:
        import random
        rng = random.Random(7)
        _CASES.extend(fz.make_case(rng) for _ in range(N))
    return _CASES[i]


_SLOW = {36: "adaptive_heun at the default rtol = 1e-7 through three training iterations: 34 s (run with -k 36 --runslow-programs)"}


@pytest.mark.parametrize("i", range(N))
def test_program_matches_the_reference_run(i):
    if i in _SLOW and not os.environ.get("TDEQ_TEST_SLOW_PROGRAMS"):
        pytest.skip(_SLOW[i].replace("run with -k 36 --runslow-programs", "TDEQ_TEST_SLOW_PROGRAMS=1 runs it"))
    case = _case(i)
    assert f"{case['kind'].__name__} {case['api']} {case['method']} {case['kw']}" == str(Z[f"p{i}_desc"]), \
        "the generator no longer produces the program the fixture was made from"
    threads = torch.get_num_threads()
    torch.set_num_threads(1)        # as the fixture was made (the states are tiny: ATen would not split them anyway)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            import torchdiffeq_amd as tda
            log = fz.run(tda, case)
    finally:
        torch.set_num_threads(threads)
    assert len(log) == int(Z[f"p{i}_n"]), (log[-1], str(Z[f"p{i}_desc"]))
    exact_grads = case["api"] == "adjoint"
    for j, (name, value) in enumerate(log):
        assert name == str(Z[f"p{i}_{j}_name"])
        if f"p{i}_{j}_none" in Z:
            assert value is None, name
            continue
        want = Z[f"p{i}_{j}_val"]
        if name == "raised":
            assert str(value)[:80] == str(want)[:80]
        elif name == "nfe":
            assert value == int(want), name
        elif exact_grads or name in ("sol", "loss"):
            got = value.numpy()
            assert got.dtype == want.dtype and got.shape == want.shape, name
            assert np.array_equal(got, want, equal_nan=True), (name, float(np.abs(got - want).max()))
        else:
            tol = 1e-11 if case["w"].dtype == torch.float64 else 1e-3
            scale = float(np.abs(want).max()) + 1e-300
            assert float(np.abs(value.numpy() - want).max()) <= tol * scale, name
