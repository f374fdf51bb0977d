"""dopri8 on small states against the reference's accepted-step sequences (tests/golden/dopri8_small.npz, generated from the
imported reference by make_golden.py `gen_dopri8_small`) — r06, VERDICT r05 weak 2 / item 6a.

The first accepted step of these solves is the initial-step heuristic's (torchdiffeq/_impl/misc.py:36-77): it matches the
reference to rounding.  Its 13-stage error row (rk_common.py:89) cancels to rounding noise, so the SECOND step size depends
on how the row sums are associated: the reference uses ATen's `torch.sum` order, the kernels add the non-zero products left
to right (`tools/dopri8_row_order.py` → profiles/r06_dopri8_row_order.json shows that only re-associating
EVERY row would remove the residue, which the carried partial sums rule out).  What is pinned here is therefore a bound,
with the measured values in the assertion messages: a regression of the row arithmetic or the controller moves them."""
import numpy as np
import pytest
import torch

import torchdiffeq_amd as tda
from _cases import DOPRI8_SMALL_CASES, StatFunc, T, dopri8_small_field, load, rel_err

# measured r06, the worst case over the three fields — oracle-backed host logic on the CPU: dt 0.28 / 6.8e-3, solution
# 1.2e-4 / 1.1e-8 (fp32 / fp64); HIP path on the MI355X (func evaluated by the GPU): dt 0.33 / 7.9e-2, solution 3.0e-4 /
# 2.6e-7.  Bounds = roughly twice the larger of the two.
BOUNDS = {"f32": dict(first_dt=2e-6, dt=0.60, y=6e-4), "f64": dict(first_dt=1e-12, dt=0.15, y=5e-7)}


@pytest.mark.parametrize("dname", ["f32", "f64"])
@pytest.mark.parametrize("name", sorted(DOPRI8_SMALL_CASES))
def test_dopri8_small_state_step_sequence_is_within_the_measured_bounds(dev, name, dname):
    z = load("dopri8_small.npz")
    key = f"d8_{name}_{dname}"
    kind = DOPRI8_SMALL_CASES[name][0]
    W, y0, t = T(z[f"{key}_W"], dev), T(z[f"{key}_y0"], dev), T(z[f"{key}_t"], dev)
    rtol, atol = (1e-5, 1e-7) if dname == "f32" else (1e-9, 1e-11)
    f = StatFunc(dopri8_small_field(kind, W))
    with torch.no_grad():
        y = tda.odeint(f, y0, t, rtol=rtol, atol=atol, method="dopri8")
    want = z[f"{key}_accept_dt"]
    b = BOUNDS[dname]
    # the heuristic's first step: no row sum involved
    assert abs(f.accept[0] - want[0]) <= b["first_dt"] * want[0], (f.accept[0], want[0])
    # same number of accepted steps, or one more / fewer where a step near the end splits
    assert abs(len(f.accept) - len(want)) <= 1, (f.accept, list(want))
    m = min(len(f.accept), len(want)) - 1          # (the last step is clipped by the output time)
    worst = max(abs(a - w) / w for a, w in zip(f.accept[:m], want[:m]))
    err = rel_err(y, z[f"{key}_y"])
    print(f"{key}[{dev}]: steps {len(f.accept)}/{len(want)}, worst dt difference {worst:.3e}, solution {err:.3e}")
    assert worst <= b["dt"], (worst, f.accept, list(want))
    assert err <= b["y"], err
