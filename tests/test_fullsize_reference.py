"""BASELINE.json's configurations at FULL size against the REFERENCE ITSELF: tests/golden/fullsize_*.npz hold what
rtqichen/torchdiffeq v0.2.5 produced on the very same inputs (tests/golden/make_golden_fullsize.py: accepted /
rejected step sequences through its callbacks, evaluation counts, sample rows of the solution and of dL/dy0, every
parameter gradient).  BASELINE.json's parity metric — max rel-err vs reference odeint — is evaluated on the sample
rows (rows 0..31, every 8th — 12.5 % of the batch, r04 —, the last 32), normalised by the reference's max|y| over all rows.

Tolerances (stated per test): fp32 at rtol 1e-7 sits on the rounding-noise floor — the reference differs from
ITSELF by 3.9e-6 there when only its CPU thread count changes — so 1e-5 is the bound, as in
BASELINE.json; evaluation counts equal; step sizes to 1 % where the error estimate is above the fp32 rounding
floor, wider (stated and explained at the assert) where it is not.

GPU tests; `TDEQ_FULLSIZE_CPU=1` additionally runs them in the build container with the CPU oracle substituted for
the kernels (host-logic check before spending GPU minutes; takes minutes)."""
import os

import numpy as np
import pytest
import torch

import torchdiffeq_amd as tda
import _fullsize as fs

_CPU_TOO = os.environ.get("TDEQ_FULLSIZE_CPU") == "1"


@pytest.fixture(params=[pytest.param("cpu", marks=pytest.mark.skipif(not _CPU_TOO, reason="TDEQ_FULLSIZE_CPU=1 only")),
                        pytest.param("cuda", marks=pytest.mark.gpu)])
def device(request, monkeypatch, oracle_kernels):
    if request.param == "cpu":
        from torchdiffeq_amd import _native
        monkeypatch.setattr(_native, "get_kernels", lambda device, dtype=None: oracle_kernels)
    return torch.device(request.param)


def _solve_linear(case, B, D, dtype, method, device, with_callbacks, options=None):
    z = fs.load(case)
    A, y0 = fs.linear_problem(B, D, dtype)
    assert np.array_equal(y0[:4].numpy(), z["y0_rows"]) and np.array_equal(A[:2].numpy(), z["A_rows"]), \
        "the seeded inputs differ from the ones the reference was run on"
    At = A.T.contiguous().to(device)
    nfe = [0]

    def field(t, y):
        nfe[0] += 1
        return y @ At
    rec = fs.Recorder(field) if with_callbacks else None
    t = torch.tensor([0.0, 1.0], dtype=torch.float64 if dtype == torch.float64 else torch.float32, device=device)
    rtol, atol = [float(v) for v in z["tol"]]
    with torch.no_grad():
        y = tda.odeint(field, y0.to(device), t, rtol=rtol, atol=atol, method=method, options=options)
    return z, y[-1], nfe[0], rec


@pytest.mark.parametrize("case,B", [("cfg2", 65536), ("cfg2_shard", 8192)])
def test_cfg2_vs_reference(case, B, device):
    """cfg2 (and its 1/8 shard): dopri5, dy/dt = Ay, fp32, rtol 1e-7 / atol 1e-9.  Reference: NFE 68, 11 accepted."""
    # the default path (look-ahead controller on the device), then the callback-driven host loop for the steps
    z, y_end, nfe, _ = _solve_linear(case, B, 128, torch.float32, "dopri5", device, with_callbacks=False)
    err = fs.sample_rel_err(y_end[torch.from_numpy(z["rows"]).to(device)], z["y_end_rows"], z["y_end_absmax"])
    assert err < 1e-5, err
    assert nfe == int(z["nfe"])
    z, y_end2, nfe2, rec = _solve_linear(case, B, 128, torch.float32, "dopri5", device, with_callbacks=True)
    assert nfe2 == int(z["nfe"])
    # step sizes: 3 % here, not 1 % — at rtol 1e-7 the fp32 error estimate of the early (small) steps is rounding
    # noise, so the ratio fed to the controller depends on the summation order (the reference itself moves its dt by
    # 0.3 % when only its thread count changes; measured against this fixture: <= 1.6 %, re-converging)
    ok, msg = fs.steps_match(rec.acc, z["accepted"], rel=3e-2)
    assert ok, msg
    assert len(rec.rej) == len(z["rejected"])
    # look-ahead and host-driven loops take the same decisions: same solution to fp32 rounding of dt
    assert fs.sample_rel_err(y_end2, y_end, z["y_end_absmax"]) < 1e-6


def test_cfg4_vs_reference(device):
    """cfg4: dopri8, 16384 x 512 fp64, rtol 1e-9 / atol 1e-11.  Reference: NFE 67 = 2 + 13*5, 5 accepted.
    dopri8's embedded error estimate is a 9-term cancelling sum; at the first (heuristic, tiny) step it is pure
    rounding noise — ATen's blocked summation order vs the kernels' left-to-right order give different noise, so
    the second step size differs; from there the sequences re-converge (docs/LAB_NOTEBOOK.md §8).  Solution bound: a fraction of the solve's own
    distance from the closed form (both 4.0e-7)."""
    z, y_end, nfe, _ = _solve_linear("cfg4", 16384, 512, torch.float64, "dopri8", device, with_callbacks=False)
    err = fs.sample_rel_err(y_end[torch.from_numpy(z["rows"]).to(device)], z["y_end_rows"], z["y_end_absmax"])
    # measured on the MI355X: 1.7e-8 between the two 5-step solutions — each is 4.0e-7 from the closed form (the
    # accuracy rtol 1e-9 buys on this problem with dopri8's five large steps), so the two agree 20x closer than either
    # is to the truth; they differ by the step sizes the noise-driven second step leads to.
    assert err < 1e-7, err
    # the reference's 5 trial steps, or one more (the noise-driven second step size overshoots and is rejected once);
    # the exact companion with a fixed first step (test_cfg4_with_a_fixed_first_step_equals_the_reference) pins the rest
    assert int(z["nfe"]) == 67 and nfe in (67, 80), nfe


def _run_cfg3(case, rows, device, with_callbacks, fp64_field=False, shared_f64_module=False, f64_state=False):
    z = fs.load(case)
    field, y0 = fs.cfg3_problem(rows)
    if f64_state:
        field, y0 = field.double(), y0.double()
    assert np.array_equal(y0[:4].numpy(), z["y0_rows"])
    for i, p in enumerate(field.net.parameters()):
        assert np.array_equal(p.detach().numpy(), z[f"p{i}"]), "layer initialisation differs from the reference run"
    field = field.to(device)
    if shared_f64_module:
        field = fs.F64MLPField(field.net)       # the module the `*_f64field` fixtures were made with
    if fp64_field:
        net64 = field.net.double()

        class F64(torch.nn.Module):
            """The same MLP evaluated in fp64 and rounded to fp32: a field with (almost) no rounding noise."""

            def __init__(self):
                super().__init__()
                self.net, self.nfe = net64, 0

            def forward(self, t, y):
                self.nfe += 1
                return self.net(y.double()).float()
        field = F64()
    rec = fs.Recorder(field) if with_callbacks else None
    x = y0.to(device).requires_grad_(True)
    t = torch.tensor([0.0, 1.0], device=device, dtype=y0.dtype)
    y = tda.odeint_adjoint(field, x, t, rtol=1e-5, atol=1e-7, method="dopri5")
    nfe_fwd, field.nfe = field.nfe, 0
    y[-1].pow(2).sum().backward()
    return z, field, x, y[-1].detach(), nfe_fwd, field.nfe, rec


@pytest.mark.parametrize("case,rows", [("cfg3", None), ("cfg3_shard", slice(0, 8192))])
def test_cfg3_adjoint_vs_reference(case, rows, device):
    """cfg3 (and its 1/8 shard): odeint_adjoint, MLP 64-256-256-64, fp32, rtol 1e-5 / atol 1e-7, loss sum(y(1)^2).
    Reference (CPU, 1 thread): forward NFE 20 (3 accepted), backward NFE 74 (12 accepted; shard: 68, 11 accepted).

    Solution, dL/dy0 and every parameter gradient must match the reference; the forward solve step for step.  The
    BACKWARD solve's step sequence is a different matter, and the reason is measured, not assumed
    (profiles/r02_cfg3_steps.json, tools/cfg3_steps.py): it starts from a tiny heuristic step (1.8e-6 / 1.4e-5: the
    parameter adjoints start at zero) and grows it by 0.9 / ratio^(1/5) per step, where `ratio` — the fp32 error
    estimate over tolerance, ~1e-3 — is nothing but the rounding noise of the field's own arithmetic (the MLP and its
    VJP).  Same solver, same inputs, only the field's arithmetic changed:
        field evaluated in fp64, rounded to fp32   growth ~5x per step    backward NFE 62 (shard 56)
        fp32 on the CPU (the reference; this package's host logic on the CPU oracle: also 74 / 68)   ~4x    NFE 74 (68)
        fp32 on the MI355X (hipBLASLt GEMMs, device tanh)                  ~3x    NFE 86 (74)
    and the reference alone moves its step sizes by up to 30 % when only its CPU thread count changes.  In fp64 every
    one of these steps grows by exactly ifactor = 10.  So: no rejected step, the first step equal (it comes from the
    initial-step heuristic, before any noise), at most three accepted steps more or fewer than the reference, and the
    fp64-evaluated field takes FEWER evaluations than the reference — the ordering above."""
    z, field, x, y_end, nfe_fwd, nfe_bwd, _ = _run_cfg3(case, rows, device, with_callbacks=False)
    idx = torch.from_numpy(z["rows"]).to(device)
    assert fs.sample_rel_err(y_end[idx], z["y_end_rows"], z["y_end_absmax"]) < 1e-5
    assert fs.sample_rel_err(x.grad[idx], z["grad_y0_rows"], z["grad_y0_absmax"]) < 1e-4
    for i, p in enumerate(field.net.parameters()):
        ref = torch.from_numpy(z[f"grad_p{i}"])
        assert float((p.grad.cpu() - ref).abs().max() / ref.abs().max()) < 1e-4, i
    assert nfe_fwd == int(z["nfe_fwd"])
    # measured on the MI355X (r02, r03 driver lines): +12 evaluations on the full batch, +6 on the shard — two / one
    # more accepted steps on the noise-limited growth phase (docstring); one trial step (6 evaluations) of slack
    measured = {"cfg3": 12, "cfg3_shard": 6}[case]
    assert measured - 6 <= nfe_bwd - int(z["nfe_bwd"]) <= measured + 6, (nfe_bwd, int(z["nfe_bwd"]))
    z, field, x, _, nfe_fwd2, nfe_bwd2, rec = _run_cfg3(case, rows, device, with_callbacks=True)
    assert (nfe_fwd2, nfe_bwd2) == (nfe_fwd, nfe_bwd)        # callbacks (host-driven loop) change nothing
    ok, msg = fs.steps_match(rec.acc, z["accepted"])
    assert ok, "forward: " + msg
    assert len(rec.rej) == 0 and len(rec.rej_adj) == 0
    ref_adj = z["accepted_adjoint"]
    assert abs(len(rec.acc_adj) - len(ref_adj)) <= 3, (len(rec.acc_adj), len(ref_adj))      # measured: +2 / +1
    ok, msg = fs.steps_match(rec.acc_adj[:1], ref_adj[:1], rel=1e-4)
    assert ok, "backward, first step: " + msg
    ok, msg = fs.steps_match(rec.acc_adj[:2], ref_adj[:2], rel=0.2)
    assert ok, "backward, second step: " + msg
    if device.type == "cuda":
        *_, nfe_fwd64, nfe_bwd64, _ = _run_cfg3(case, rows, device, with_callbacks=False, fp64_field=True)
        # the fp64-evaluated field is the least noisy of the three: fewest evaluations (measured 62 < 74 <= 86 and
        # 56 < 68 <= 74; the middle inequality depends on the BLAS library's kernel choice and is not asserted)
        assert nfe_fwd64 == nfe_fwd and nfe_bwd64 < int(z["nfe_bwd"]) and nfe_bwd64 < nfe_bwd, \
            (nfe_bwd64, int(z["nfe_bwd"]), nfe_bwd)


@pytest.mark.parametrize("trace", ["closed", "autograd"])
def test_cfg5_cnf_adjoint_vs_reference(trace, device):
    """cfg5 as BASELINE.json writes it: CNF (examples/cnf.py model at its seeded init), state (z[32768,2],
    logp[32768,1]), t: 10 -> 0, dopri5 + adjoint, rtol = atol = 1e-5, loss mean(logp) - sum(z^2)/100.  The reference
    run used the example's own per-dimension autograd trace; "autograd" restates that loop, "closed" is the same
    quantity in closed form.  Reference: forward NFE 44 (5 accepted, 2 rejected), backward 11 accepted."""
    z = fs.load("cfg5")
    z0, logp0 = fs.cfg5_problem()
    assert np.array_equal(z0[:4].numpy(), z["z0_rows"])
    cnf = fs.ExampleCNF([z[f"p{i}"] for i in range(6)], trace=trace).to(device)
    rec = fs.Recorder(cnf)
    x = z0.to(device).requires_grad_(True)
    t = torch.tensor([10.0, 0.0], device=device)
    zt, lp = tda.odeint_adjoint(cnf, (x, logp0.to(device)), t, atol=1e-5, rtol=1e-5, method="dopri5")
    nfe_fwd, cnf.nfe = cnf.nfe, 0
    loss = lp[-1].mean() - zt[-1].pow(2).sum() / 100
    loss.backward()
    idx = torch.from_numpy(z["rows"]).to(device)
    # bounds = 5x what the driver's own bench run measured on the MI355X (BENCH_r03 `configs.cfg5`: z 8.7e-7, logp 2.8e-5,
    # loss 9.7e-7, dL/dz0 8.0e-7, parameter gradients <= 1.7e-6) — rtol = atol = 1e-5 is what the configuration asks for
    assert fs.sample_rel_err(zt[-1][idx], z["z_end_rows"], z["z_end_absmax"]) < 5e-6
    assert fs.sample_rel_err(lp[-1][idx], z["logp_end_rows"], z["logp_end_absmax"]) < 1.5e-4
    assert abs(float(loss.detach()) - float(z["loss"])) < 5e-6 * abs(float(z["loss"]))
    assert fs.sample_rel_err(x.grad[idx], z["grad_z0_rows"], z["grad_z0_absmax"]) < 5e-6
    for i, p in enumerate(cnf.parameters()):
        ref = torch.from_numpy(z[f"grad_p{i}"])
        assert float((p.grad.cpu() - ref).abs().max() / ref.abs().max()) < 1e-5, i
    assert nfe_fwd == int(z["nfe_fwd"]), nfe_fwd
    ok, msg = fs.steps_match(rec.acc, z["accepted"])
    assert ok, "forward: " + msg
    assert len(rec.rej) == len(z["rejected"])
    # Backward: the same 11 accepted steps; sizes to 35 % and at most one extra rejected trial step (6 evaluations) —
    # the fp32 error estimate of this backward solve sits on the rounding floor (see test_cfg3_adjoint_vs_reference),
    # and with the closed-form trace one trial step lands at an error ratio within rounding of 1 and flips to a reject.
    ok, msg = fs.steps_match(rec.acc_adj, z["accepted_adjoint"], rel=0.35)
    assert ok, "backward: " + msg
    assert len(rec.rej_adj) <= len(z["rejected_adjoint"]) + 1
    assert cnf.nfe == int(z["nfe_bwd"]) + 6 * (len(rec.rej_adj) - len(z["rejected_adjoint"])), cnf.nfe


def test_cfg5_hutchinson_trace_variant(device):
    """The benchmark-side Hutchinson variant of cfg5 (one fixed Rademacher probe).  In two dimensions e^T J e =
    tr J + (J01 + J10) e0 e1, so the estimator is unbiased but not exact: z(t1) does not depend on the trace at all
    and must match the reference; logp differs by the probe's cross term (bounded here, not pinned)."""
    z = fs.load("cfg5")
    z0, logp0 = fs.cfg5_problem()
    cnf = fs.ExampleCNF([z[f"p{i}"] for i in range(6)], trace="hutchinson").to(device)
    x = z0.to(device).requires_grad_(True)
    t = torch.tensor([10.0, 0.0], device=device)
    zt, lp = tda.odeint_adjoint(cnf, (x, logp0.to(device)), t, atol=1e-5, rtol=1e-5, method="dopri5")
    (lp[-1].mean() - zt[-1].pow(2).sum() / 100).backward()
    idx = torch.from_numpy(z["rows"]).to(device)
    assert fs.sample_rel_err(zt[-1][idx], z["z_end_rows"], z["z_end_absmax"]) < 1e-3
    assert torch.isfinite(lp[-1]).all() and torch.isfinite(x.grad).all()
    assert all(torch.isfinite(p.grad).all() for p in cnf.parameters())
    assert abs(float(lp[-1].mean()) - float(torch.from_numpy(z["logp_end_rows"]).mean())) < 0.5


# ---------------------------------------------------------------------------------------------------------------------
# Exact companions of the three noise-limited comparisons above (r03).  Each of the wide tolerances above — cfg3's
# backward evaluation count (+-18), cfg4's count (+-13) and solution bound (1e-7), cfg5's backward step sizes (35 %) —
# is explained by rounding noise that the two libraries do not share (the field's own fp32 arithmetic; dopri8's first,
# heuristic step).  Here the noise is removed AT ITS SOURCE, identically on both sides, and the comparison becomes
# exact: the reference was run on the very same noise-free module / with the same fixed first step
# (make_golden_fullsize.py: cfg3_f64field, cfg3_shard_f64field, cfg4_first_step, cfg5_f64field).
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case,rows", [("cfg3_f64field", None), ("cfg3_shard_f64field", slice(0, 8192))])
def test_cfg3_backward_solve_under_equal_field_noise(case, rows, device):
    """cfg3 with the MLP evaluated in fp64 on BOTH sides (tests/_fullsize.F64MLPField): evaluation counts EQUAL forward
    and backward (the +-18 of the fp32-field test is the field's noise, as claimed there), the same number of accepted
    steps, none rejected, the first backward step — which comes from the initial-step heuristic, before any noise — to
    1e-6.  What is left in the later step sizes is the fp32 rounding of the SOLVER's own stage sums, which the two
    libraries order differently (ATen's blocked sum vs left to right): the backward error estimate is ~2e-4 of the
    tolerance (each step grows ~5x), i.e. rounding residue, and a 12 % spread of dt is a 50 % spread of that residue
    (measured: <= 12.2 % on the shard, <= 16.4 % at full size; 25 % asserted).  The companion below removes that too."""
    z, field, x, y_end, nfe_fwd, nfe_bwd, rec = _run_cfg3(case, rows, device, with_callbacks=True, shared_f64_module=True)
    assert (nfe_fwd, nfe_bwd) == (int(z["nfe_fwd"]), int(z["nfe_bwd"])), (nfe_fwd, nfe_bwd)
    ok, msg = fs.steps_match(rec.acc, z["accepted"], rel=3e-2)
    assert ok, "forward: " + msg
    ok, msg = fs.steps_match(rec.acc_adj[:1], z["accepted_adjoint"][:1], rel=1e-6)
    assert ok, "backward, first step: " + msg
    ok, msg = fs.steps_match(rec.acc_adj, z["accepted_adjoint"], rel=0.25)
    assert ok, "backward: " + msg
    assert len(rec.rej) == len(z["rejected"]) and len(rec.rej_adj) == len(z["rejected_adjoint"])
    idx = torch.from_numpy(z["rows"]).to(device)
    assert fs.sample_rel_err(y_end[idx], z["y_end_rows"], z["y_end_absmax"]) < 1e-5
    assert fs.sample_rel_err(x.grad[idx], z["grad_y0_rows"], z["grad_y0_absmax"]) < 1e-5
    for i, p in enumerate(field.net.parameters()):
        ref = torch.from_numpy(z[f"grad_p{i}"])
        assert float((p.grad.cpu() - ref).abs().max() / ref.abs().max()) < 2e-5, i
    # and the look-ahead path (no callbacks) takes the same number of evaluations
    *_, nfe_fwd2, nfe_bwd2, _ = _run_cfg3(case, rows, device, with_callbacks=False, shared_f64_module=True)
    assert (nfe_fwd2, nfe_bwd2) == (nfe_fwd, nfe_bwd)


def test_cfg3_shard_in_fp64_is_step_for_step_the_reference(device):
    """The same shard, same numbers, carried in fp64 (state, parameters, times; rtol 1e-5 / atol 1e-7 as cfg3): no fp32
    rounding anywhere, and the adjoint's backward solve matches the reference STEP FOR STEP — every accepted step size,
    forward and backward, to 1e-6, equal evaluation counts, gradients to 1e-9."""
    z, field, x, y_end, nfe_fwd, nfe_bwd, rec = _run_cfg3("cfg3_shard_f64state", slice(0, 8192), device,
                                                          with_callbacks=True, f64_state=True)
    assert (nfe_fwd, nfe_bwd) == (int(z["nfe_fwd"]), int(z["nfe_bwd"])), (nfe_fwd, nfe_bwd)
    for mine, ref, what in ((rec.acc, z["accepted"], "forward"), (rec.acc_adj, z["accepted_adjoint"], "backward")):
        ok, msg = fs.steps_match(mine, ref, rel=1e-6)
        assert ok, what + ": " + msg
    assert len(rec.rej) == len(z["rejected"]) and len(rec.rej_adj) == len(z["rejected_adjoint"])
    idx = torch.from_numpy(z["rows"]).to(device)
    assert fs.sample_rel_err(y_end[idx], z["y_end_rows"], z["y_end_absmax"]) < 1e-9
    assert fs.sample_rel_err(x.grad[idx], z["grad_y0_rows"], z["grad_y0_absmax"]) < 1e-9
    for i, p in enumerate(field.net.parameters()):
        ref = torch.from_numpy(z[f"grad_p{i}"])
        assert float((p.grad.cpu() - ref).abs().max() / ref.abs().max()) < 1e-9, i


def test_cfg4_with_a_fixed_first_step_equals_the_reference(device):
    """cfg4 with options={'first_step': 0.1}: no heuristic first step, hence no step whose error estimate is pure
    rounding noise — evaluation count EQUAL (1 + 13 per trial step), every step size to 1 %, solution to 1e-9."""
    z, y_end, nfe, rec = _solve_linear("cfg4_first_step", 16384, 512, torch.float64, "dopri8", device, with_callbacks=True,
                                       options=dict(first_step=0.1))
    assert nfe == int(z["nfe"]), (nfe, int(z["nfe"]))
    ok, msg = fs.steps_match(rec.acc, z["accepted"], rel=1e-2)
    assert ok, msg
    assert len(rec.rej) == len(z["rejected"])
    err = fs.sample_rel_err(y_end[torch.from_numpy(z["rows"]).to(device)], z["y_end_rows"], z["y_end_absmax"])
    assert err < 1e-9, err
    z, y_end2, nfe2, _ = _solve_linear("cfg4_first_step", 16384, 512, torch.float64, "dopri8", device, with_callbacks=False,
                                       options=dict(first_step=0.1))
    assert nfe2 == nfe and fs.sample_rel_err(y_end2, y_end, z["y_end_absmax"]) < 1e-12


def test_cfg5_backward_solve_under_equal_field_noise(device):
    """cfg5's flow evaluated in fp64 on both sides (ExampleCNF(trace='closed', f64=True) with cfg5's parameters):
    equal evaluation counts and rejections, forward steps to 1 %, backward steps to 5 %, gradients to 1e-5."""
    z, p = fs.load("cfg5_f64field"), fs.load("cfg5")
    z0, logp0 = fs.cfg5_problem()
    cnf = fs.ExampleCNF([p[f"p{i}"] for i in range(6)], trace="closed", f64=True).to(device)
    rec = fs.Recorder(cnf)
    x = z0.to(device).requires_grad_(True)
    t = torch.tensor([10.0, 0.0], device=device)
    zt, lp = tda.odeint_adjoint(cnf, (x, logp0.to(device)), t, atol=1e-5, rtol=1e-5, method="dopri5")
    nfe_fwd, cnf.nfe = cnf.nfe, 0
    loss = lp[-1].mean() - zt[-1].pow(2).sum() / 100
    loss.backward()
    assert (nfe_fwd, cnf.nfe) == (int(z["nfe_fwd"]), int(z["nfe_bwd"])), (nfe_fwd, cnf.nfe)
    # forward to 1 %; backward to 5 % (measured 2.4 %: what is left is the solver's own fp32 stage-sum rounding, see
    # test_cfg3_backward_solve_under_equal_field_noise) — against 35 % with the fp32 field
    for mine, ref, what, rel in ((rec.acc, z["accepted"], "forward", 1e-2),
                                 (rec.acc_adj, z["accepted_adjoint"], "backward", 5e-2)):
        ok, msg = fs.steps_match(mine, ref, rel=rel)
        assert ok, what + ": " + msg
    assert len(rec.rej) == len(z["rejected"]) and len(rec.rej_adj) == len(z["rejected_adjoint"])
    idx = torch.from_numpy(z["rows"]).to(device)
    errs = {"z": fs.sample_rel_err(zt[-1][idx], z["z_end_rows"], z["z_end_absmax"]),
            "logp": fs.sample_rel_err(lp[-1][idx], z["logp_end_rows"], z["logp_end_absmax"]),
            "grad_z0": fs.sample_rel_err(x.grad[idx], z["grad_z0_rows"], z["grad_z0_absmax"])}
    for i, q in enumerate(cnf.parameters()):
        ref = torch.from_numpy(z[f"grad_p{i}"])
        errs[f"grad_p{i}"] = float((q.grad.cpu() - ref).abs().max() / ref.abs().max())
    # (fp32 state at rtol = atol = 1e-5: the two solutions differ by the step-size spread above; measured on the CPU
    # host logic: z 3e-6, logp 2.1e-5, gradients <= 4e-5)
    assert errs["z"] < 2e-5 and errs["logp"] < 1e-4 and all(v < 2e-4 for k, v in errs.items() if k.startswith("grad")), errs


# ---------------------------------------------------------------------------------------------------------------------
# r04: the reference's arithmetic ON THE SAME DEVICE.  The package's torch-op host path evaluates the reference's own
# expressions with ATen and is bit-identical to the reference on the CPU (tests/test_hostpath.py,
# tools/fuzz_vs_reference.py hostexact); forced onto the cuda state it is what the reference would compute on this GPU —
# same field arithmetic on both sides, ALL rows compared.
# ---------------------------------------------------------------------------------------------------------------------
class _ReferenceArithmetic:
    def __enter__(self):
        from torchdiffeq_amd import _fallback, _native
        self._native, self._orig = _native, _native.get_kernels
        host = _fallback.HostKernels()
        _native.get_kernels = lambda d, dt=None: host
        return self

    def __exit__(self, *exc):
        self._native.get_kernels = self._orig


@pytest.mark.gpu
@pytest.mark.parametrize("case,B,D,dtype,method,bound", [("cfg2", 65536, 128, torch.float32, "dopri5", 1e-5),
                                                         ("cfg4", 16384, 512, torch.float64, "dopri8", 1e-9)])
def test_linear_configs_vs_the_references_arithmetic_on_the_same_gpu(case, B, D, dtype, method, bound):
    dev = torch.device("cuda")
    z, y_hip, nfe_hip, _ = _solve_linear(case, B, D, dtype, method, dev, with_callbacks=False)
    with _ReferenceArithmetic():
        _, y_ref, nfe_ref, _ = _solve_linear(case, B, D, dtype, method, dev, with_callbacks=False)
    assert nfe_hip == nfe_ref == int(z["nfe"])
    err = float((y_hip - y_ref).abs().max() / y_ref.abs().max())        # every row (measured: 4.8e-6 / 9.8e-11)
    assert err < bound, err


@pytest.mark.gpu
def test_cfg3_adjoint_vs_the_references_arithmetic_on_the_same_gpu():
    """With the field's arithmetic equal on both sides the backward solve takes the same number of evaluations (measured
    86 = 86; the CPU reference's 74 belong to the CPU's GEMMs and tanh, not to the solver) and the gradients agree to a
    few 1e-6."""
    dev = torch.device("cuda")
    z, field, x, y_end, nfe_fwd, nfe_bwd, _ = _run_cfg3("cfg3", None, dev, with_callbacks=False)
    g_hip = [x.grad.clone()] + [p.grad.clone() for p in field.net.parameters()]
    with _ReferenceArithmetic():
        _, field2, x2, y_end2, nfe_fwd2, nfe_bwd2, _ = _run_cfg3("cfg3", None, dev, with_callbacks=False)
    g_ref = [x2.grad] + [p.grad for p in field2.net.parameters()]
    assert nfe_fwd == nfe_fwd2 == int(z["nfe_fwd"])
    assert abs(nfe_bwd - nfe_bwd2) <= 6, (nfe_bwd, nfe_bwd2)                 # at most one trial step apart (measured: equal)
    assert float((y_end - y_end2).abs().max() / y_end2.abs().max()) < 1e-5
    for a, b in zip(g_hip, g_ref):
        assert float((a - b).abs().max() / b.abs().max()) < 5e-5
