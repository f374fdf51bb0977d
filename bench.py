#!/usr/bin/env python
"""bench.py — BASELINE.json's headline metric on the MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload linear|adjoint] [--scaling weak|strong]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches itself as N ranks (one process per GPU,
`python -m torch.distributed.run ... bench.py`, rendezvous on 127.0.0.1, backend nccl = RCCL); under
`torch.distributed.run` it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* itself.  Rank r owns GPU r.
More ranks than visible GPUs is REFUSED (one JSON error line, exit status 2) — never a silent fallback to another
backend; `TDEQ_DIST_BACKEND=gloo` asks for the multi-rank control flow with ranks sharing devices by name.  Before
anything is timed every N > 1 run all-reduces a ones tensor through the backend ON THE DEVICE and gathers each rank's GPU
identity: the line carries `backend`, `rccl_ranks` (what RCCL summed over; must equal N) and `comm.devices`; a census
that does not match ends the run with an error line and status 3.

Workloads
  linear   (default) BASELINE.json configs[1] / SURVEY.md §8d cfg2 — the configuration the metric is quoted on:
           dopri5, dy/dt = A y, state 65536 x 128 fp32 per GPU, rtol 1e-7 / atol 1e-9, synthetic seeded data.
           One "step" = one dopri5 trial step of the adaptive solver = 6 RK stages: 6 `stage_combine` launches
           interleaved with 6 evaluations of the field (a 65536x128x128 GEMM run by PyTorch-ROCm), one fused
           `error_norm` launch, one read-back and the host step controller.  State resident in HBM before the timed
           region.  Weak scaling by default (every rank integrates its own 65536-row shard with its own accept/reject
           loop, no data-path collective — SURVEY.md §8e); `--scaling strong` splits the ONE 65536-row batch.
  adjoint  BASELINE.json configs[2] / cfg3 — odeint_adjoint, MLP 64-256-256-64, 65536 x 64 fp32, rtol 1e-5 /
           atol 1e-7, loss sum(y(1)^2); rows sharded over the ranks, parameter gradients summed by ONE all-reduce
           (RCCL over xGMI) at the end of backward (torchdiffeq_amd.dist.odeint_adjoint_sharded).  One "step" = one
           forward + backward pass; value = RK stages (forward + backward solves) per second.

Timing: W warm-up steps, then 5 blocks of exactly K steps each, every block bracketed by a barrier +
torch.cuda.synchronize(); per block the MAX over ranks; `value` / `ms_per_step` are the MEDIAN block (min / max in
`blocks`).

Extra objects in the JSON line (linear workload):
  roofline      dominant launch = the step's 7-words-per-element stage combine (234.9 MB): tableau row 5 launched row by
                row (5 k_j + y0 read, y_i written: stage_combine_kernel<float, 5>) or, with carried partial sums
                (tableaus.carry_plan — on for dopri5 at this size), row 4 + the prefix of row 5's sum
                (stage_combine_multi_kernel<float, 4>, 2 outputs) — `kernel` / `carry_plan` say which;
                `achieved`/`frac` = IN SITU: its launches inside the timed region, each stamped by the dispatch itself
                (tdeq_stage_combine[_multi]_timed -> hipExtLaunchKernelGGL start/stop events) — every launch when
                K <= 50, every 4th otherwise; the stage tensors were written by `func` just before, so part of the
                reads is served by the 256 MiB Infinity Cache.  `cold` = the same kernel on 4 rotating buffer sets
                (940 MB > 256 MiB), i.e. every byte from HBM (`cold_row_by_row_kernel`: r02's kernel, for continuity).
                `traffic` = HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/
                (`traffic_source`; counters cannot be read from inside this process).
  solver_only   the step's solver kernels alone, back to back on the last step's stage tensors.
  shard_regime  (N = 1) what one GPU does on a 1/8 shard of the batch — the per-rank work of an 8-GPU strong-scaling
                run: 8192 x 128 linear trial steps (host-driven, look-ahead, hip_graph) and the 8192 x 64 adjoint pass.
  configs       (N = 1) the other BASELINE.json configurations, bounded (<= 15 s): cfg4 (dopri8 fp64 16384 x 512: odeint
                ms, NFE vs the reference's, rel-err vs tests/golden/fullsize_cfg4.npz, roofline of its dominant launch in
                situ + cold), cfg5 (CNF + adjoint: forward / backward ms eager and with captured steps, rel-errs vs
                fullsize_cfg5.npz), cfg1 (rk4 spiral: GPU eager / captured, and on the CPU as BASELINE writes it,
                with the bit-equality flag vs the reference's trajectory).
  weak, strong, lockstep, adjoint   (N > 1) the same ranks on both scaling regimes of cfg2, in lock-step mode, and on
                cfg3 (strong and weak; forward / backward ms, evaluation counts, `allreduce.{calls, bytes, ms}`);
                `extras_s` = seconds each took; a watchdog (TDEQ_BENCH_EXTRAS_TIMEOUT, 240 s) prints the line marked
                `extras_timed_out` + `extras_hung_in` if one of them hangs.
  comm, backend, rccl_ranks   what the collective backend connected (see above).
  rel_err_vs_reference  full odeint(t=[0,1]) of cfg2 vs the REFERENCE's own result on these inputs
                (tests/golden/fullsize_cfg2.npz, sample rows; rank 0), next to rel_err vs the closed form.
  reference_style_eager_gpu   the reference's eager op sequence restated in stock PyTorch on the same GPU
                (oracle/eager_torch_port.py — pinned op for op to the reference's step sequences,
                tests/test_oracle_golden.py::test_eager_torch_port_is_the_reference_op_for_op).
  same_device_reference, adjoint_same_device_reference, configs.cfg4.same_device_reference   (N = 1) the same solves
                once more on this GPU with the REFERENCE's arithmetic — the package's torch-op host path (the reference's own
                expressions evaluated by ATen; bit-identical to the reference on the CPU, tools/fuzz_vs_reference.py
                hostexact) forced onto the cuda state: evaluation counts of both, step-size differences, rel-err over ALL rows
                / gradients.  "max rel-err vs reference odeint" with the field's arithmetic equal on both sides.
  cpu_baseline  the CPU oracle (a port: the reference itself cannot travel to the GPU box) on a bounded sample, on
                rank 0 at N = 1, with BASELINE.md's figure for the real reference on 8 cores beside it.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BATCH, DIM = 65536, 128
RTOL, ATOL = 1e-7, 1e-9
ADJ_BATCH, ADJ_DIM = 65536, 64
HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_BLOCKS = 5


# ---------------------------------------------------------------------------------------------------
# problems
# ---------------------------------------------------------------------------------------------------
def make_problem(device, seed_offset=0, rows=None):
    """cfg2 synthetic inputs (per-rank seed offset for the weak-scaling shards; `rows` = strong-scaling shard of
    rank 0's batch)."""
    g = torch.Generator().manual_seed(0)
    G = torch.randn(DIM, DIM, generator=g, dtype=torch.float64) / DIM ** 0.5
    A = (0.5 * (G - G.T) - 0.1 * torch.eye(DIM, dtype=torch.float64)).float()
    # rank 0 draws y0 from the same generator right after A (exactly the survey's cfg2 inputs);
    # other ranks draw their own 65536 rows from generator seed = rank.
    gy = g if seed_offset == 0 else torch.Generator().manual_seed(seed_offset)
    y0 = torch.randn(BATCH, DIM, generator=gy, dtype=torch.float64).float()
    if rows is not None:
        y0 = y0[rows].contiguous()
    return A.to(device), y0.to(device)


def dist_sync(world):
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world, device):
    if world == 1:
        return x
    v = torch.tensor([x], device=device, dtype=torch.float64)
    torch.distributed.all_reduce(v, op=torch.distributed.ReduceOp.MAX)
    return float(v.item())


def timed_blocks(step_fn, steps, warmup, world, device, n_blocks=N_BLOCKS):
    """W warm-up calls, then n_blocks blocks of exactly `steps` calls; per block barrier + synchronize on both sides
    and the max over ranks.  Returns the per-block seconds."""
    for _ in range(warmup):
        step_fn()
    blocks = []
    timed_blocks.local = []          # this rank's own block times, before the barrier (skew diagnosis at N > 1)
    for _ in range(n_blocks):
        dist_sync(world)
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        torch.cuda.synchronize()
        timed_blocks.local.append(time.perf_counter() - t0)
        if world > 1:
            torch.distributed.barrier()
        blocks.append(max_over_ranks(time.perf_counter() - t0, world, device))
    return blocks


class EventTimedKernels:
    """Forwards to HipKernels; while `armed`, the step's dominant stage-combine launch — the one that moves 7 words per
    element (234.9 MB at cfg2) — goes through the `_timed` entry point, whose dispatch stamps a pair of HIP events with
    its own begin / end timestamps (hipExtLaunchKernelGGL).  Row by row that launch is tableau row 5 (5 stages + y0
    read, y_5 written: stage_combine_kernel<float, 5>); with carried partial sums (tableaus.carry_plan, on for dopri5 at
    this size) it is row 4's launch (4 stages + y0 read; y_4 and the prefix of row 5's sum written:
    stage_combine_multi_kernel<float, 4>) — same bytes."""

    def __init__(self, inner, dominant_terms, n_events, every):
        self._seen = 0
        self._inner = inner
        self._nt = dominant_terms
        self.every = every         # an event-stamped dispatch costs a few microseconds of pipeline
        self.armed = False
        self.events = []
        self.kernel = None
        # events are created (and recorded once: torch creates the hipEvent_t lazily) before the timed region
        self._pool = []
        for _ in range(n_events):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            self._pool.append((e0, e1))

    def __getattr__(self, name):
        return getattr(self._inner, name)

    def _take(self):
        self._seen += 1
        if self._seen % self.every == 0 and self._pool:
            ev = self._pool.pop()
            self.events.append(ev)
            return ev
        return None

    def stage_combine(self, out, y0, ks, coefs, dt):
        if self.armed and len(ks) == self._nt:
            ev = self._take()
            if ev is not None:
                self.kernel = f"stage_combine_kernel<float, {self._nt}, 1, true>"
                self._inner.stage_combine_timed(out, y0, ks, coefs, dt, ev[0], ev[1])
                return
        self._inner.stage_combine(out, y0, ks, coefs, dt)

    def stage_combine_multi(self, outs, rows, y0, acc_in, ks, dt, events=None):
        if self.armed and acc_in is None and len(ks) + 1 + len(outs) == self._nt + 2:
            ev = self._take()
            if ev is not None:
                self.kernel = f"stage_combine_multi_kernel<float, {len(ks)}, true> ({len(outs)} outputs)"
                return self._inner.stage_combine_multi(outs, rows, y0, acc_in, ks, dt, events=ev)
        return self._inner.stage_combine_multi(outs, rows, y0, acc_in, ks, dt)


# ---------------------------------------------------------------------------------------------------
# baselines
# ---------------------------------------------------------------------------------------------------
def cpu_baseline(max_seconds=20.0):
    """Oracle (port of the reference algorithm) timed on this host's cores on a bounded sample."""
    from oracle import reference_solver as orc
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    A, y0 = make_problem("cpu")
    A, y0 = A.numpy(), y0.numpy()
    ops = orc.COps()
    field = orc.LinearField(A)
    solver = orc.AdaptiveRK(lambda tt, y: field.f(tt, y.reshape(BATCH, DIM)).reshape(-1), y0.reshape(-1),
                            orc.tableau("dopri5"), RTOL, ATOL, ops=ops)
    solver.before_integrate(0.0)
    solver.adaptive_step()                      # warm-up step (page faults, thread pool)
    steps, t0 = 0, time.perf_counter()
    while steps < 40 and time.perf_counter() - t0 < max_seconds:
        solver.adaptive_step()
        steps += 1
    dt = time.perf_counter() - t0
    torch_cpu = None
    try:
        # the reference's OWN op sequence on torch's CPU path (oracle/eager_torch_port.py: stage-minor k, ~220 ATen
        # ops per trial step, 0-dim tensor scalars) on this host's cores — the closest thing to "the reference on
        # this box's CPU" that can travel; bounded to a few trial steps
        from oracle import eager_torch_port as ep
        # torch's CPU ops on 33 MB tensors are fastest at 8-16 threads on the GPU box's 256-thread host (measured,
        # seconds per trial step: 8: 0.58, 16: 0.55, 32: 0.75, 64: 1.2, 128: 2.2, 256: 11.8)
        tthreads = min(cores, 16)
        torch.set_num_threads(tthreads)
        At_cpu = torch.from_numpy(A).T.contiguous()
        eager = ep.EagerAdaptiveRK(lambda tt, y: y @ At_cpu, torch.from_numpy(y0), 0.0, 0.05, RTOL, ATOL, "dopri5")
        with torch.no_grad():
            eager.adaptive_step()
            n_e, t1 = 0, time.perf_counter()
            while n_e < 6 and time.perf_counter() - t1 < 12.0:
                eager.adaptive_step()
                n_e += 1
            dte = time.perf_counter() - t1
        torch_cpu = {"value": 6 * n_e / dte, "unit": "RK-stages/s", "cores": tthreads, "kind": "port",
                     "sample": f"{n_e} dopri5 trial steps of the same workload through the reference's eager op "
                               f"sequence on torch CPU tensors ({tthreads} threads: the fastest setting on this "
                               f"host), {dte:.1f} s"}
    except Exception as exc:
        torch_cpu = {"error": repr(exc)}
    return {"value": 6 * steps / dt, "unit": "RK-stages/s", "cores": cores, "kind": "port",
            "reference_op_sequence_on_torch_cpu": torch_cpu,
            "sample": f"{steps} dopri5 trial steps ({6 * steps} RK stages) of the same 65536x128 fp32 workload, "
                      f"oracle/rk_oracle.c with OpenMP on {cores} threads + numpy GEMM, {dt:.1f} s",
            "why_a_port": "the reference is a Python package mounted only in the build container (/root/reference); it "
                          "does not exist on the GPU box, so the CPU leg there is the committed restatement of its "
                          "algorithm (oracle/, pinned to the reference's outputs by tests/test_oracle_golden.py)",
            "reference_8core": {"value": 3.08, "unit": "RK-stages/s", "cores": 8, "kind": "reference",
                                "source": "BASELINE.md §2: rtqichen/torchdiffeq v0.2.5 itself, this workload at full "
                                          "size, torch CPU on the build container's 8-core Xeon (21.44 s for NFE 68)"}}


def eager_gpu_baseline(field, y0, first_step, steps=12):
    """The reference's own way of running this workload on a GPU — stock eager PyTorch-ROCm ops with 0-dim device
    tensors for the time-like scalars (oracle/eager_torch_port.py, a restatement: the reference itself cannot
    travel to the GPU box) — timed on the same MI355X, same state, same field."""
    from oracle import eager_torch_port as ep
    solver = ep.EagerAdaptiveRK(field, y0, 0.0, first_step, RTOL, ATOL, "dopri5")
    for _ in range(3):
        solver.adaptive_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        solver.adaptive_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": 6 * steps / dt, "unit": "RK-stages/s", "ms_per_step": 1e3 * dt / steps, "kind": "port",
            "sample": f"{steps} dopri5 trial steps of the same 65536x128 fp32 workload through oracle/eager_torch_port.py "
                      "(the reference's eager op sequence: stage-minor k tensor, ~220 ATen ops and ~19 host syncs "
                      "per trial step) on this GPU"}


# ---------------------------------------------------------------------------------------------------
# linear workload (cfg2)
# ---------------------------------------------------------------------------------------------------
class reference_arithmetic:
    """Context: every solve inside runs on the package's torch-op host path — the reference's own expressions evaluated by
    ATen (bit-identical to the reference on the CPU) — whatever device the state lives on.  For the same-device
    comparisons of the bench line only (the product selects its backend by the state alone)."""

    def __enter__(self):
        from torchdiffeq_amd import _fallback, _native
        self._native, self._orig = _native, _native.get_kernels
        host = _fallback.HostKernels()
        _native.get_kernels = lambda d, dt=None: host
        return self

    def __exit__(self, *exc):
        self._native.get_kernels = self._orig


def same_device_reference(field, y0, device):
    """BASELINE.json's "max rel-err vs reference odeint" with the reference's arithmetic ON THE SAME GPU.  The reference
    itself is a Python package that does not exist on the GPU box; the package's torch-op host path evaluates the
    reference's own expressions with ATen — `torch.sum` over the dense tableau row, `abs().pow(2).mean().sqrt()` — and is
    bit-identical to the reference wherever the two can be run side by side (the CPU: tools/fuzz_vs_reference.py hostexact,
    750 cases).  Forced onto the cuda state it is what `torchdiffeq.odeint(..., device='cuda')` computes: same func (the
    same hipBLASLt GEMM), ATen's GPU reductions instead of the HIP kernels.  Reported: both solves' evaluation counts and
    accepted step sizes, and max|y - y_ref| / max|y_ref| over ALL rows."""
    import torchdiffeq_amd as tda
    from torchdiffeq_amd import _fallback, _native
    t = torch.tensor([0.0, 1.0], device=device)
    runs = {}
    for name in ("hip", "reference_arithmetic"):
        steps, nfe = [], [0]

        class F(torch.nn.Module):
            def forward(self, t_, y_):
                nfe[0] += 1
                return field(t_, y_)

            def callback_accept_step(self, t0, y, dt):
                steps.append(float(dt))
        orig = _native.get_kernels
        if name != "hip":
            host = _fallback.HostKernels()
            _native.get_kernels = lambda d, dt=None: host
        try:
            with torch.no_grad():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                y = tda.odeint(F(), y0, t, rtol=RTOL, atol=ATOL, method="dopri5")[-1]
                torch.cuda.synchronize()
                runs[name] = (y, nfe[0], steps, time.perf_counter() - t0)
        finally:
            _native.get_kernels = orig
    (y, nfe, st, wall), (yr, nfe_r, st_r, wall_r) = runs["hip"], runs["reference_arithmetic"]
    n = min(len(st), len(st_r))
    return {"what": "odeint(t=[0,1]) of this workload twice on this GPU: HIP kernels vs the reference's own torch expressions "
                    "(torchdiffeq_amd._fallback.HostKernels forced onto the cuda state; bit-identical to the reference on the CPU)",
            "rel_err_all_rows": float((y - yr).abs().max() / yr.abs().max()),
            "nfe": nfe, "reference_arithmetic_nfe": nfe_r, "accepted": len(st), "reference_arithmetic_accepted": len(st_r),
            "max_rel_step_size_difference": max((abs(a - b) / b for a, b in zip(st[:n], st_r[:n])), default=None),
            "wall_s": wall, "reference_arithmetic_wall_s": wall_r,
            "note": "the HIP solve runs with callbacks here (host-driven loop), like its twin"}


def make_stepper(field, y0, hip_graph=False, lookahead=None, dist_sync=None, rtol=None, atol=None):
    """A Dopri5Solver in the middle of a long solve (no output time ahead — where the look-ahead first stage
    applies), ready for `_trial_step()` calls."""
    from torchdiffeq_amd.misc import OdeFunc, StateLayout, rms_norm
    from torchdiffeq_amd.solvers import Dopri5Solver
    layout = StateLayout([y0.shape], False)
    func = OdeFunc(field, layout, 1.0, y0.dtype, y0.device)
    prev = os.environ.get("TDEQ_LOOKAHEAD")
    if lookahead is not None:
        os.environ["TDEQ_LOOKAHEAD"] = "1" if lookahead else "0"
    try:
        solver = Dopri5Solver(func=func, y0=y0.reshape(-1), rtol=RTOL if rtol is None else rtol,
                              atol=ATOL if atol is None else atol, norm=rms_norm, hip_graph=hip_graph, dist_sync=dist_sync)
    finally:
        if lookahead is not None:
            if prev is None:
                os.environ.pop("TDEQ_LOOKAHEAD", None)
            else:
                os.environ["TDEQ_LOOKAHEAD"] = prev
    solver._before_integrate([0.0])
    solver._t_end = float("inf")
    return solver


def time_steps(solver, steps, warmup, world, device, n_blocks=N_BLOCKS):
    with torch.no_grad():
        blocks = timed_blocks(solver._trial_step, steps, warmup, world, device, n_blocks)
        if solver._g is not None:
            torch.cuda.synchronize()
            solver._g.release()
    return blocks


def block_stats(blocks, steps):
    ms = sorted(1e3 * b / steps for b in blocks)
    return {"median": statistics.median(ms), "min": ms[0], "max": ms[-1], "n_blocks": len(ms), "steps_per_block": steps}


def solver_only_rate(solver, device):
    """the 6 stage_combine launches + error_norm (+ finalize) of one dopri5 step on the k tensors
    of the last timed step, no func, HIP events around REPS back-to-back passes."""
    rec = solver._dense
    kern = solver.kernels._inner if hasattr(solver.kernels, "_inner") else solver.kernels
    ks, y0s = rec.k, rec.y0
    outs = [torch.empty_like(y0s) for _ in range(2)]
    REPS = 30
    fuse = solver._fuse
    epart = torch.empty_like(y0s)
    last = len(solver._beta) - 1
    la = bool(solver._lookahead and fuse is not None)
    tnext = torch.empty(len(solver._beta), dtype=y0s.dtype, device=device)
    carry = solver._carry
    carry_bufs = {t: torch.empty_like(y0s) for op in (carry.ops if carry is not None else ()) if op is not None
                  for t in op.targets[1:]}

    def one_pass():
        # exactly the solver's launch sequence for one trial step, minus func (and, without look-ahead, the
        # stage-time fill)
        held = {}
        for i, row in enumerate(solver._beta):
            op = carry.ops[i] if (carry is not None and i > 0) else None
            if i == 0 and la:
                kern.stage_combine_sel(outs[0], rec.y1, ks[-1], y0s, ks[0], row.coef[0], solver.plan)
            elif carry is not None and i > 0 and op is None:
                held.pop(i)                           # finished by an earlier launch of the plan
            elif op is not None and not (len(op.targets) == 1 and not op.continues) and \
                    not (op.targets == (i, last + 1) and i == last and not op.continues):
                bufs = [outs[i & 1]] + [carry_bufs[t] for t in op.targets[1:]]
                kern.stage_combine_multi(bufs, op.spec, y0s, held.pop(i) if op.continues else None,
                                         [ks[j] for j in op.idx], rec.dt_signed)
                for t, b in zip(op.targets[1:], bufs[1:]):
                    held[t] = b
            elif i == last and fuse is not None:
                kern.stage_combine_err(outs[i & 1], epart, y0s, [ks[j] for j in row.idx], row.coef, fuse[0],
                                       rec.dt_signed)
            else:
                kern.stage_combine(outs[i & 1], y0s, [ks[j] for j in row.idx], row.coef, rec.dt_signed)
        if la:
            solver._ctrl.t0, solver._ctrl.dt = rec.t0, rec.t1 - rec.t0
            kern.error_norm_partial_ctrl(solver.plan, epart, y0s, rec.y1, [ks[j] for j in fuse[1]], fuse[2],
                                         rec.dt_signed, solver._ctrl, tnext)
        elif fuse is not None:
            kern.error_norm_partial(solver.plan, epart, y0s, rec.y1, [ks[j] for j in fuse[1]], fuse[2], rec.dt_signed)
        else:
            kern.error_norm(solver.plan, y0s, rec.y1, [ks[j] for j in solver._c_err.idx], solver._c_err.coef,
                            rec.dt_signed)
    for _ in range(3):
        one_pass()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        one_pass()
    e1.record()
    torch.cuda.synchronize()
    solver.plan.expect = ()     # the passes' results are not read back
    t_step = e0.elapsed_time(e1) * 1e-3 / REPS
    n = y0s.numel()
    # SURVEY.md §8(d) counts 32 + 8 = 40 words per element for a dopri5 step; the end-of-step fusion moves
    # 37 (3+4+5+6+7+8 for the six combines, 4 for the norm) — both rates are reported.
    moved = (carry.words if carry is not None else (37 if fuse is not None else 40)) * n * 4
    survey = 40 * n * 4
    return {"stages_per_s": 6 / t_step, "us_per_step": 1e6 * t_step,
            "bytes_moved_per_step": moved, "GBps_moved": moved / t_step / 1e9,
            "frac_of_hbm_peak_moved": moved / t_step / 1e9 / HBM_PEAK_GBPS,
            "survey_algorithmic_bytes_per_step": survey, "GBps_survey_bytes": survey / t_step / 1e9,
            "note": "the solver's own launch sequence for one dopri5 trial step (stage_combine_sel + 4 "
                    "stage_combine + stage_combine_err + error_norm_partial + controller finalize) back "
                    "to back, no func; the 7 k tensors (235 MB) + y0/y1 fit the 256 MiB Infinity Cache "
                    "only partly"}


def cold_dominant_kernel(kern, n, device, sets=4, launches=24, carried=False):
    """The dominant launch (7 words per element) on `sets` rotating buffer sets whose total size exceeds the 256 MiB
    Infinity Cache several times: every read comes from HBM.  Timed per launch by the dispatch's own start/stop
    events.  carried=False: stage_combine_kernel<float, 5> (5 stages + y0 -> y); carried=True: the two-output launch of
    the carried-partial-sum plan (4 stages + y0 -> y, prefix)."""
    g = torch.Generator(device="cpu").manual_seed(1)
    nk = 4 if carried else 5
    bufs = []
    for _ in range(sets):
        y0 = torch.randn(n, generator=g).to(device)
        ks = [torch.randn(n, generator=g).to(device) for _ in range(nk)]
        bufs.append((y0, ks, [torch.empty(n, device=device) for _ in range(2 if carried else 1)]))
    coefs = (0.1, -0.2, 0.3, 0.25, -0.15)[:nk]
    spec = ((coefs, (1 << nk) - 1, True), (tuple(-c for c in coefs), (1 << nk) - 1, False))

    def launch(b, ev=None):
        y0, ks, outs = b
        if carried:
            kern.stage_combine_multi(outs, spec, y0, None, ks, 0.1, events=ev)
        elif ev is None:
            kern.stage_combine(outs[0], y0, ks, coefs, 0.1)
        else:
            kern.stage_combine_timed(outs[0], y0, ks, coefs, 0.1, ev[0], ev[1])
    for b in bufs:                              # first touch
        launch(b)
    torch.cuda.synchronize()
    evs = []
    for i in range(launches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        launch(bufs[i % sets], (e0, e1))
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in evs]
    avg = sum(ms) / len(ms)
    bytes_per_launch = 7 * n * 4
    ach = bytes_per_launch / (avg * 1e-3) / 1e9
    return {"achieved": ach, "frac": ach / HBM_PEAK_GBPS, "avg_launch_ms": avg, "launches_timed": len(ms),
            "buffer_sets": sets, "working_set_bytes": sets * 7 * n * 4,
            "kernel": "stage_combine_multi_kernel<float, 4, true> (2 outputs)" if carried
                      else "stage_combine_kernel<float, 5, 1, true>",
            "note": "same kernel, rotating buffer sets larger than the 256 MiB Infinity Cache: all reads from HBM"}


def pmc_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary that has it
    (tools/profile_gpu.sh; counters cannot be read from inside this process)."""
    import re
    prefix = "tdeq::" + (kernel_name or "").split(" (")[0].rsplit(", true>", 1)[0]
    paths = [q for q in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json"))
             if re.fullmatch(r"r\d+[a-z]?_pmc_hbm\.json", os.path.basename(q))]      # profiles of THIS command only
    for pmc_path in sorted(paths, reverse=True):
        try:
            kernels = json.load(open(pmc_path))["kernels"]
            hit = [v for k, v in kernels.items() if k.startswith(prefix)]
            if hit:
                return hit[0]["hbm_bytes_per_launch"], os.path.relpath(pmc_path, ROOT)
        except Exception:
            continue
    return None, None


def reference_rel_err(y_end):
    """max rel-err of y(1) vs the REFERENCE's result on the same inputs (sample rows of tests/golden/fullsize_cfg2.npz)."""
    import _fullsize as fs
    z = fs.load("cfg2")
    rows = torch.from_numpy(z["rows"]).to(y_end.device)
    return fs.sample_rel_err(y_end[rows], z["y_end_rows"], z["y_end_absmax"]), int(z["nfe"])


def kernel_breakdown(step_fn, steps, is_solver=lambda name: "tdeq::" in name):
    """Where the GPU time of `steps` calls of step_fn goes: every device kernel's own duration (roctracer activity
    records through torch.profiler — also the kernels a hipGraph replay launches), split into the package's kernels
    (`tdeq::*`) and everything else (= the user's func: GEMMs, activation / autograd kernels, copies), next to the wall
    time of the same calls.  `floor_us` = the sum of kernel durations per call: what a call would cost if not a single
    microsecond were lost between dispatches — with an opaque func the lower bound of this launch sequence."""
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    solver_us = func_us = 0.0
    n_solver = n_func = 0
    per_kernel = {}
    for ev in prof.events():
        if str(getattr(ev, "device_type", "")).upper().endswith("CPU"):
            continue
        dur = float(getattr(ev, "device_time", None) or getattr(ev, "cuda_time", None) or 0.0)
        if dur <= 0.0:
            continue
        name = ev.name
        k = per_kernel.setdefault(name, [0, 0.0])
        k[0] += 1
        k[1] += dur
        if is_solver(name):
            solver_us += dur
            n_solver += 1
        else:
            func_us += dur
            n_func += 1
    top = sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:8]
    return {"calls": steps, "wall_us_per_call_profiled": 1e6 * wall / steps,
            "solver_kernel_us": solver_us / steps, "func_kernel_us": func_us / steps,
            "floor_us": (solver_us + func_us) / steps,
            "dispatches_per_call": (n_solver + n_func) / steps, "solver_dispatches_per_call": n_solver / steps,
            "func_dispatches_per_call": n_func / steps,
            "top_kernels": {n[:80]: {"calls_per_call": c / steps, "avg_us": t / c} for n, (c, t) in top},
            "source": "torch.profiler (roctracer kernel activity records); the profiled wall is slower than the timed "
                      "blocks' — use `wall_us` from the timed blocks next to `floor_us`"}


def strong_breakdown(step_fn, steps, wall_ms, world, rank):
    """The per-rank answer to 'launch gaps or kernel floor?' for a strong-scaling shard: gathers every rank's
    kernel_breakdown and derives the floor of the whole job (the slowest rank's)."""
    try:
        mine = kernel_breakdown(step_fn, steps)
    except Exception as exc:          # the profiler is evidence, never a reason to lose the line
        mine = {"error": repr(exc)}
    mine["rank"] = rank
    ranks = [mine]
    if world > 1:
        ranks = [None] * world
        torch.distributed.all_gather_object(ranks, mine)
    floors = [r["floor_us"] for r in ranks if "floor_us" in r]
    for r in ranks:         # the per-kernel table of rank 0 is enough in the line
        if r.get("rank", 0) != 0:
            r.pop("top_kernels", None)
            r.pop("source", None)
    out = {"per_rank": ranks, "wall_ms_per_step": wall_ms}
    if floors:
        floor_ms = max(floors) * 1e-3
        out.update({"floor_ms": floor_ms, "gap_ms": max(0.0, wall_ms - floor_ms),
                    "gap_is": "wall - floor, clamped at 0: the end stamp of a graph node and the start stamp of the next "
                              "overlap by a fraction of a microsecond, so a gap-free replay can sum to slightly MORE than its wall",
                    "floor_is": "max over ranks of (sum of kernel durations per step): func (opaque to the package) + "
                                "solver kernels, zero time between dispatches",
                    "func_floor_ms": max(r["func_kernel_us"] for r in ranks if "floor_us" in r) * 1e-3,
                    "solver_floor_ms": max(r["solver_kernel_us"] for r in ranks if "floor_us" in r) * 1e-3})
    return out


def shard_regime_linear(device, steps=100, warmup=20):
    """One GPU on the 8192 x 128 shard (1/8 of cfg2): ms per trial step on the three step paths."""
    A, y0 = make_problem(device, rows=slice(0, BATCH // 8))
    At = A.T.contiguous()
    field = lambda t, y: y @ At
    out = {"state": f"{BATCH // 8} x {DIM} fp32 (1/8 of cfg2)", "steps_per_block": steps}
    for name, kw in (("host_driven", dict(lookahead=False)), ("lookahead", dict(lookahead=True)),
                     ("hip_graph", dict(hip_graph=True)), ("auto", dict(hip_graph="auto"))):
        try:
            solver = make_stepper(field, y0, **kw)
            # ("auto": first sight of this func -> eager until solvers._AUTO_CAPTURE_AFTER_STEPS trial steps, then captured)
            blocks = time_steps(solver, steps, warmup if name != "auto" else warmup + 110, 1, device, n_blocks=3)
            st = block_stats(blocks, steps)
            out[name] = {"ms_per_step": st["median"], "min": st["min"], "max": st["max"],
                         "stages_per_s_of_the_shard": 6e3 / st["median"]}
            if name == "auto":
                out[name]["replaying"] = solver._g is not None
            if name == "hip_graph":
                solver = make_stepper(field, y0, **kw)
                with torch.no_grad():
                    for _ in range(warmup):
                        solver._trial_step()
                    out[name]["breakdown"] = strong_breakdown(solver._trial_step, 50, st["median"], 1, 0)
                    if solver._g is not None:
                        torch.cuda.synchronize()
                        solver._g.release()
        except Exception as exc:
            out[name] = {"error": repr(exc)}
    return out


def strong_scaling_prediction(device, full_ms, steps=100, warmup=20):
    """What ONE GPU does on the shard an N-GPU strong-scaling run of cfg2 gives it (N = 2, 4, 8: 32768 / 16384 / 8192 rows),
    on the step path `bench.py --gpus N` takes for that shard (captured steps up to 2^21 elements, the look-ahead path
    above).  No data-path collective exists, so the run's step time is the slowest shard's: 6 / this = the predicted
    `value`, full-size step / this = the predicted speed-up over N = 1 — the curve the 8-GPU node will be measured against."""
    out = {}
    for n_gpus in (2, 4, 8):
        rows = BATCH // n_gpus
        A, y0 = make_problem(device, rows=slice(0, rows))
        At = A.T.contiguous()
        graph = y0.numel() <= (1 << 21)
        try:
            solver = make_stepper(lambda t, y: y @ At, y0, hip_graph=graph)
            st = block_stats(time_steps(solver, steps, warmup, 1, device, n_blocks=3), steps)
            out[str(n_gpus)] = {"rows_per_gpu": rows, "elements": y0.numel(), "step_path": "hip_graph" if graph else "lookahead",
                                "ms_per_step": st["median"], "predicted_value_RK_stages_per_s": 6e3 / st["median"],
                                "predicted_speedup_over_n1": full_ms / st["median"]}
        except Exception as exc:
            out[str(n_gpus)] = {"error": repr(exc)}
        del solver
        torch.cuda.empty_cache()
    return out


def run_linear(args, rank, world, device, parity=True):
    import torchdiffeq_amd as tda
    strong = args.scaling == "strong" and world > 1
    if strong:
        from torchdiffeq_amd.dist import shard_rows
        A, y0 = make_problem(device, rows=shard_rows(BATCH, rank, world))
    else:
        A, y0 = make_problem(device, seed_offset=rank)
    At = A.T.contiguous()
    field = lambda t, y: y @ At
    n = y0.numel()

    # ---- parity at full size: whole odeint vs the closed form and vs the reference's own result ----
    rel_err = rel_err_ref = ref_nfe = odeint_wall = None
    nfe = [0]

    def counted(t, y):
        nfe[0] += 1
        return y @ At
    with torch.no_grad():
        t_wall = time.perf_counter()
        if not parity:          # (second regime of a multi-rank run: only the timed steps)
            y_end = None
        else:
            y_end = tda.odeint(counted, y0, torch.tensor([0.0, 1.0], device=device), rtol=RTOL, atol=ATOL,
                               method="dopri5")[-1]
        torch.cuda.synchronize()
        if parity:
            odeint_wall = time.perf_counter() - t_wall
            exact = y0.double() @ torch.linalg.matrix_exp(A.double()).T
            rel_err = float((y_end.double() - exact).abs().max() / exact.abs().max())
            if rank == 0 and not strong:
                try:
                    rel_err_ref, ref_nfe = reference_rel_err(y_end)
                except Exception as exc:
                    rel_err_ref = repr(exc)
            del exact

    # ---- timed region ----
    auto_graph = strong and n <= (1 << 21) and os.environ.get("TDEQ_BENCH_GRAPH", "1") != "0"
    solver = make_stepper(field, y0, hip_graph=auto_graph)
    every = 1 if args.steps <= 50 else 4
    timed = EventTimedKernels(solver.kernels, dominant_terms=5, n_events=N_BLOCKS * args.steps // every + 1, every=every)
    solver.kernels = timed
    solver.ops.k = timed        # the elementwise kernels are issued through solver.ops
    with torch.no_grad():
        for _ in range(args.warmup):
            solver._trial_step()
        timed.armed = True
        blocks = timed_blocks(solver._trial_step, args.steps, 0, world, device)
        timed.armed = False
    st = block_stats(blocks, args.steps)
    ms_per_step = st["median"]
    per_rank = None
    if world > 1:       # every rank's own median block (no barrier inside): shows a straggler GPU, if any
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, 1e3 * statistics.median(timed_blocks.local) / args.steps)
    # weak: every rank did its own stages; strong: a stage of the global batch is done when every shard's is
    value = 6e3 / ms_per_step * (1 if strong else world)

    breakdown = None
    if os.environ.get("TDEQ_BENCH_BREAKDOWN") and world == 1:
        with torch.no_grad():
            breakdown = kernel_breakdown(solver._trial_step, min(20, args.steps))
    if strong:
        # launch gaps or kernel floor?  (every rank profiles its own shard's steps; collective: all ranks call this)
        with torch.no_grad():
            breakdown = strong_breakdown(solver._trial_step, min(50, args.steps), ms_per_step, world, rank)

    out = None
    if rank == 0:
        kernel_ms = [a.elapsed_time(b) for a, b in timed.events]
        avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        bytes_per_launch = 7 * n * 4
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if kernel_ms else None
        traffic, traffic_src = pmc_traffic(timed.kernel) if n == BATCH * DIM else (None, None)
        out = {
            "metric": "dopri5 RK-stages/sec at batch=65536x dim=128 (end-to-end adaptive trial steps incl. func, "
                      "error norm, read-back and host controller)",
            "value": value, "unit": "RK-stages/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: dopri5 adaptive, linear ODE dy/dt=Ay, batch=65536 x "
                                   "dim=128 fp32 " + ("in total, rows split over the GPUs" if strong else "per GPU")
                                   + ", rtol=1e-7 atol=1e-9",
                       "global_batch": BATCH if strong else BATCH * world, "rows_per_gpu": y0.shape[0], "dim": DIM,
                       "parallelism": f"batch-sharded x{world}", "accepted": solver.n_accepted,
                       "rejected": solver.n_rejected, "lookahead": bool(solver._lookahead),
                       "hip_graph": bool(solver.hip_graph),
                       "backend": torch.distributed.get_backend() if world > 1 else None},
            "blocks": {"ms_per_step": st, "value_is": "median block", "per_rank_ms_per_step": per_rank},
            "rel_err_vs_reference": rel_err_ref,
            "rel_err_vs_reference_definition": "max|y - y_ref| over the sample rows / max|y_ref| of odeint(t=[0,1]) at "
                                               "full size, y_ref = rtqichen/torchdiffeq v0.2.5 on the same inputs "
                                               "(tests/golden/fullsize_cfg2.npz; rank 0's shard = SURVEY cfg2)",
            "nfe": nfe[0], "reference_nfe": ref_nfe,
            "rel_err": rel_err,
            "rel_err_definition": "max|y - y_exact| / max|y_exact| vs the closed form y0 @ expm(A)^T (the reference's "
                                  "own fp32 result scores 2.2-2.8e-6 on this)",
            "odeint_t01_wall_s": odeint_wall,
        }
        if breakdown is not None:
            out["breakdown"] = breakdown
        if n != BATCH * DIM:
            # a strong-scaling shard: its steps are hipGraph replays (no dispatch-stamped events), so the dominant
            # launch's duration comes from the breakdown's kernel-activity records of rank 0
            top = ((breakdown or {}).get("per_rank") or [{}])[0].get("top_kernels", {})
            hit = [(k, v["avg_us"]) for k, v in top.items() if "tdeq::" in k and "<float, 5, true, false>" in k] or \
                  [(k, v["avg_us"]) for k, v in top.items() if "tdeq::stage_combine_kernel<float, 5" in k
                   or "tdeq::stage_combine_multi_kernel<float, 4" in k]
            if kernel_ms:       # an eager shard (> 2^21 elements): the dispatch-stamped events of the timed blocks
                hit = [(timed.kernel, 1e3 * avg_ms)]
            if hit:
                name, avg_us = hit[0]
                rec = {"avg_us": avg_us}
                bytes_per_launch = 7 * n * 4
                ach = bytes_per_launch / (rec["avg_us"] * 1e-6) / 1e9
                out["roofline"] = {
                    "bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": bytes_per_launch,
                    "avg_launch_ms": rec["avg_us"] * 1e-3, "traffic": None,
                    "timing": ("HIP events stamped by the dispatch itself, timed blocks, rank 0" if kernel_ms else
                               "roctracer kernel-activity records of the replayed graph nodes (torch.profiler), rank 0"),
                    "note": f"1/{world} shard: the launch's seven streams ({bytes_per_launch / 1e6:.1f} MB) fit the 256 MiB "
                            "Infinity Cache, so this is a cache rate measured against the HBM peak; the full-size kernel's "
                            "HBM figures are in the N = 1 line (`roofline.frac`, `roofline.frac_hbm_cold`)"}
        if n == BATCH * DIM:
            out["roofline"] = {
                "bound": "hbm", "kernel": timed.kernel, "achieved": achieved,
                "kernel_is": "the step's 7-words-per-element stage-combine launch (234.9 MB): row 5 launched row by row "
                             "(stage_combine_kernel<float, 5>), or row 4 + the carried prefix of row 5 under "
                             "tableaus.carry_plan (stage_combine_multi_kernel<float, 4>, 2 outputs) — `carry_plan` says which",
                "carry_plan": solver._carry is not None,
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
                "frac_is": "in situ (stage tensors freshly written by func; partly Infinity-Cache resident) — see `cold`",
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms,
                "timing": "HIP events stamped by the dispatch itself (hipExtLaunchKernelGGL start/stop) on the launch "
                          f"stream, every {'launch' if every == 1 else '4th launch'} of this kernel in the timed blocks",
                "launches_timed": len(kernel_ms), "traffic": traffic,
                "traffic_source": (traffic_src + " (rocprofv3 --pmc passes of this same command, replayed — not "
                                   "measured in this run)") if traffic_src else None}
            try:
                out["roofline"]["cold"] = cold_dominant_kernel(timed._inner, n, device, carried=solver._carry is not None)
                # against HBM alone (every byte from DRAM) — the figure to quote as "fraction of the HBM roofline"; `frac`
                # above is the same kernel where the solver runs it, with the 256 MiB Infinity Cache helping
                out["roofline"]["frac_hbm_cold"] = out["roofline"]["cold"].get("frac")
                if solver._carry is not None:       # continuity with r01 / r02: the row-by-row kernel, cold
                    out["roofline"]["cold_row_by_row_kernel"] = cold_dominant_kernel(timed._inner, n, device)
            except Exception as exc:
                out["roofline"]["cold"] = {"error": repr(exc)}
            try:
                out["solver_only"] = solver_only_rate(solver, device)
            except Exception as exc:      # never let the extra figure break the contract line
                out["solver_only"] = {"error": repr(exc)}
    if solver._g is not None:
        torch.cuda.synchronize()
        solver._g.release()
    return out, field, y0


# ---------------------------------------------------------------------------------------------------
# r05 regimes (N = 1, extras file): reduced-precision states, per-element tolerances, the func lever
# ---------------------------------------------------------------------------------------------------
def lowp_steps(dtype, backend, steps, warmup, device):
    """dopri5 trial steps of the cfg2-shaped workload with a bf16 / fp16 STATE: `backend` "hip" = the kernels of
    csrc/tdeq_kernels_lp.hpp (what a reduced-precision cuda state selects), "torch-op" = the package's torch-op host path
    forced onto the same device (what r04 ran for such states)."""
    from torchdiffeq_amd import _fallback, _native
    A, y0 = make_problem(device)
    # a pure rotation (the skew-symmetric part of cfg2's matrix): |y| stays put — with cfg2's -0.1 I the state decays below
    # atol, a 16-bit error estimate becomes exactly 0 and `ratio == 0 -> dt * ifactor` (misc.py:88) runs dt to inf
    A = (A + 0.1 * torch.eye(DIM, device=device)).to(dtype)
    y0 = y0.to(dtype)
    At = A.T.contiguous()
    orig = _native.get_kernels
    if backend == "torch-op":
        low = _fallback.LowPrecisionHostKernels()
        _native.get_kernels = lambda dev_, dt_=None: low if dt_ in (torch.bfloat16, torch.float16) else orig(dev_, dt_)
    try:
        blocks = []
        with torch.no_grad():
            for _ in range(3):          # a fresh solve per block (a 16-bit solve of this field lasts ~100 steps)
                solver = make_stepper(lambda t, y: y @ At, y0, rtol=1e-2, atol=1e-3)
                for _ in range(warmup):
                    solver._trial_step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    solver._trial_step()
                torch.cuda.synchronize()
                blocks.append((time.perf_counter() - t0) / steps)
            breakdown = None
            if backend == "hip":
                b = kernel_breakdown(solver._trial_step, 10)
                breakdown = {k: b[k] for k in ("solver_kernel_us", "func_kernel_us", "floor_us", "dispatches_per_call",
                                               "top_kernels")}
    finally:
        _native.get_kernels = orig
    ms = 1e3 * statistics.median(blocks)
    return {"backend": solver.kernels.name, "lookahead": bool(solver._lookahead), "ms_per_step": ms,
            "rk_stages_per_s": 6e3 / ms, "accepted": solver.n_accepted, "rejected": solver.n_rejected,
            "steps_timed": steps, "breakdown": breakdown}


def lowp_combine_rate(dtype, device, nt=5, n=BATCH * DIM, sets=8, launches=48):
    """The 16-bit stage combine (nt stages + y0 read, y_i written: 7 streams of 16.8 MB) on rotating buffer sets (cold)
    and on one set (warm)."""
    from torchdiffeq_amd import _native
    k = _native.get_kernels(device, dtype)
    bufs = [(torch.randn(n, device=device).to(dtype), [torch.randn(n, device=device).to(dtype) for _ in range(nt)],
             torch.empty(n, dtype=dtype, device=device)) for _ in range(sets)]
    coefs = (0.1, -0.2, 0.3, 0.25, -0.15, 0.05, 0.4)[:nt]
    for y0, ks, out in bufs:
        k.stage_combine(out, y0, ks, coefs, 0.1)
    torch.cuda.synchronize()

    def timed(rotate):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(launches):
            y0, ks, out = bufs[i % sets if rotate else 0]
            k.stage_combine(out, y0, ks, coefs, 0.1)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / launches
    nbytes = (nt + 2) * n * 2
    out = {"kernel": f"lp::map_kernel<{'BF16' if dtype == torch.bfloat16 else 'F16'}, {nt + 1}, 1, true, CombineF>",
           "algorithmic_bytes_per_launch": nbytes}
    for label, rotate in (("cold", True), ("warm", False)):
        ms = statistics.median(timed(rotate) for _ in range(5))
        out[label] = {"avg_launch_ms": ms, "GBps": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / HBM_PEAK_GBPS,
                      "buffer_sets": sets if rotate else 1}
    return out


def low_precision_regime(device):
    res = {"workload": "dopri5 trial steps, dy/dt = A y (rotation), 65536 x 128, rtol 1e-2 atol 1e-3, state in bf16 / fp16"}
    for name, dtype in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        entry = {"stage_combine": lowp_combine_rate(dtype, device)}
        try:
            hip = lowp_steps(dtype, "hip", 40, 5, device)
            ref = lowp_steps(dtype, "torch-op", 10, 2, device)
            entry.update({"hip_kernels": hip, "torch_op_host_path": ref, "speedup": ref["ms_per_step"] / hip["ms_per_step"]})
        except AssertionError as exc:
            # float16: the initial-step heuristic underflows the type's range in the reference as well ("underflow in dt
            # 0.0", tests/test_brow_golden.py) — adaptive solves of fp16 states do not start; fixed grids do
            entry["adaptive_steps"] = {"error": str(exc)}
        res[name] = entry
    return res


def vector_tolerance_regime(field, y0, device, steps=60, warmup=10):
    """cfg2 trial steps with a PER-ELEMENT rtol (an fp64 vector over the state, misc.py:80-82): the fused launch
    (tdeq_error_norm_vec: the tolerance vector is one more 8-byte stream of the norm kernel) vs the r04 route (raw error
    materialised + the scaling and the norm as fp64 torch ops) vs the scalar-tolerance step next to them."""
    rtol_vec = torch.full(y0.shape, RTOL, dtype=torch.float64, device=device)
    out = {}
    for label, kw, fused in (("scalar_tolerances", {}, None), ("vector_rtol_fused", dict(rtol=rtol_vec), True),
                             ("vector_rtol_torch_ops", dict(rtol=rtol_vec), False)):
        solver = make_stepper(field, y0, **kw)
        if fused is False:
            solver._vec_fused = None
        with torch.no_grad():
            for _ in range(warmup):
                solver._trial_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                solver._trial_step()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / steps
        out[label] = {"ms_per_step": ms, "lookahead": bool(solver._lookahead), "fused_norm": solver._vec_fused is not None}
    out["extra_ms_fused"] = out["vector_rtol_fused"]["ms_per_step"] - out["scalar_tolerances"]["ms_per_step"]
    out["extra_ms_torch_ops"] = out["vector_rtol_torch_ops"]["ms_per_step"] - out["scalar_tolerances"]["ms_per_step"]
    out["note"] = "per-element tolerances run host-driven steps (no look-ahead); the scalar line is the default path"
    return out


def tunableop_lever(args):
    """The headline workload once more in a child process with PyTorch's TunableOp switched on (the user-side lever on
    `func`: its six y @ A.T GEMMs are 45 % of the step and run at a third of the HBM rate under hipBLASLt's default
    heuristic).  Reported NEXT to the headline, never instead of it: the contract value stays the default-heuristic one."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_VERBOSE="0",
                   PYTORCH_TUNABLEOP_FILENAME=os.path.join(tmp, "tunableop_results.csv"), TDEQ_BENCH_EXTRAS_DIR=tmp,
                   TDEQ_BENCH_BREAKDOWN="1")
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps),
                                "--warmup", str(max(args.warmup, 10)), "--no-extras", "--no-cpu-baseline"], env=env,
                               capture_output=True, text=True, timeout=float(os.environ.get("TDEQ_TUNABLEOP_TIMEOUT", "150")))
        except subprocess.TimeoutExpired:
            return {"error": "tuning did not finish within the time bound"}
        took = time.perf_counter() - t0
        try:
            child = json.load(open(os.path.join(tmp, "bench_extras_n1.json")))
        except Exception:
            return {"error": "child produced no result", "stderr_tail": r.stderr[-300:]}
    bd = child.get("breakdown") or {}
    return {"env": "PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1", "ms_per_step": child.get("ms_per_step"),
            "value": child.get("value"), "func_kernel_us": bd.get("func_kernel_us"),
            "solver_kernel_us": bd.get("solver_kernel_us"), "rel_err_vs_reference": child.get("rel_err_vs_reference"),
            "nfe": child.get("nfe"), "child_wall_s": round(took, 1)}


# ---------------------------------------------------------------------------------------------------
# the other BASELINE.json configurations, bounded (N = 1 line, `configs` object)
# ---------------------------------------------------------------------------------------------------
class MultiTimedKernels:
    """Forwards to HipKernels; while armed, tdeq_stage_combine_multi launches with `n_terms` stage streams and
    `n_out` outputs are stamped by the dispatch itself (tdeq_stage_combine_multi_timed)."""

    def __init__(self, inner, n_terms, n_out, n_events):
        self._inner, self._key = inner, (n_terms, n_out)
        self.armed, self.events, self.words = False, [], None
        self._pool = []
        for _ in range(n_events):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            self._pool.append((e0, e1))

    def __getattr__(self, name):
        return getattr(self._inner, name)

    def stage_combine_multi(self, outs, rows, y0, acc_in, ks, dt, events=None):
        if self.armed and (len(ks), len(outs)) == self._key and self._pool:
            ev = self._pool.pop()
            self.events.append(ev)
            self.words = len(ks) + 1 + (0 if acc_in is None else 1) + len(outs)
            return self._inner.stage_combine_multi(outs, rows, y0, acc_in, ks, dt, events=ev)
        return self._inner.stage_combine_multi(outs, rows, y0, acc_in, ks, dt)


def cfg4_config(device):
    """configs[3]: dopri8 fp64, 16384 x 512, rtol 1e-9 / atol 1e-11 — whole odeint vs the reference's own result
    (tests/golden/fullsize_cfg4.npz) and the roofline of its dominant launch, in situ and HBM-cold."""
    import _fullsize as fs
    import torchdiffeq_amd as tda
    from torchdiffeq_amd import tableaus as tb
    from torchdiffeq_amd.misc import OdeFunc, StateLayout, rms_norm
    from torchdiffeq_amd.solvers import Dopri8Solver
    z = fs.load("cfg4")
    A, y0 = fs.linear_problem(16384, 512, torch.float64)
    At, y0 = A.T.contiguous().to(device), y0.to(device)
    rtol, atol = [float(v) for v in z["tol"]]
    t = torch.tensor([0.0, 1.0], dtype=torch.float64, device=device)
    nfe = [0]

    def field(tt, y):
        nfe[0] += 1
        return y @ At
    with torch.no_grad():
        y_end = tda.odeint(field, y0, t, rtol=rtol, atol=atol, method="dopri8")[-1]
        n_eval, nfe[0] = nfe[0], 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            tda.odeint(field, y0, t, rtol=rtol, atol=atol, method="dopri8")
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 3
    with torch.no_grad(), reference_arithmetic():
        nfe[0] = 0
        y_same = tda.odeint(field, y0, t, rtol=rtol, atol=atol, method="dopri8")[-1]
        nfe_same, nfe[0] = nfe[0], 0
    same = {"rel_err_all_rows": float((y_end - y_same).abs().max() / y_same.abs().max()), "reference_arithmetic_nfe": nfe_same,
            "what": "the reference's own torch expressions on this GPU (see `same_device_reference` of the line)"}
    del y_same
    rows = torch.from_numpy(z["rows"]).to(device)
    out = {"workload": "BASELINE.json configs[3]: dopri8 fp64, batch=16384 x dim=512, rtol=1e-9 atol=1e-11",
           "same_device_reference": same,
           "odeint_t01_ms": ms, "nfe": n_eval, "reference_nfe": int(z["nfe"]),
           "rk_stages_per_s": (n_eval - 2) / (ms * 1e-3),
           "rel_err_vs_reference": fs.sample_rel_err(y_end[rows], z["y_end_rows"], z["y_end_absmax"]),
           "words_per_element_and_step": {"row_by_row": tb.row_by_row_words(tb.DOPRI8),
                                          "carried_partial_sums": tb.carry_plan("dopri8").words}}
    # dominant launch of the planned step: row 10 of the tableau, 9 stage streams + y0 read, 4 streams written
    layout = StateLayout([y0.shape], False)
    solver = Dopri8Solver(func=OdeFunc(lambda tt, y: y @ At, layout, 1.0, y0.dtype, y0.device), y0=y0.reshape(-1),
                          rtol=rtol, atol=atol, norm=rms_norm)
    if solver._carry is not None:
        solver._before_integrate([0.0])
        solver._t_end = float("inf")
        timed = MultiTimedKernels(solver.kernels, 9, 4, 16)
        solver.kernels = timed
        with torch.no_grad():
            for _ in range(2):
                solver._trial_step()
            timed.armed = True
            for _ in range(12):
                solver._trial_step()
            timed.armed = False
        torch.cuda.synchronize()
        msk = [a.elapsed_time(b) for a, b in timed.events]
        n = y0.numel()
        if msk:
            avg = sum(msk) / len(msk)
            nbytes = timed.words * n * 8
            roof = {"bound": "hbm", "kernel": "stage_combine_multi_kernel<double, 9, true> (dopri8 row 10: 9 stages + "
                                              "y0 read; y_10, y_11, the row-12 prefix and the error prefix written)",
                    "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": avg, "launches_timed": len(msk),
                    "achieved": nbytes / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": nbytes / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "frac_is": "in situ; the launch moves 939 MB, 3.7x the 256 MiB Infinity Cache"}
            # cold: the same launch on two rotating buffer sets (2 x 939 MB)
            op = tb.carry_plan("dopri8").ops[10]
            g = torch.Generator(device="cpu").manual_seed(2)
            sets = []
            for _ in range(2):
                sets.append((torch.randn(n, generator=g, dtype=torch.float64).to(device),
                             [torch.randn(n, generator=g, dtype=torch.float64).to(device) for _ in op.idx],
                             [torch.empty(n, dtype=torch.float64, device=device) for _ in op.targets]))
            kern = timed._inner
            evs = []
            for i in range(10):
                yb, kb, ob = sets[i % 2]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                e1.record()
                kern.stage_combine_multi(ob, op.spec, yb, None, kb, 0.1, events=(e0, e1))
                evs.append((e0, e1))
            torch.cuda.synchronize()
            cold = [a.elapsed_time(b) for a, b in evs[2:]]
            cavg = sum(cold) / len(cold)
            roof["cold"] = {"avg_launch_ms": cavg, "achieved": nbytes / (cavg * 1e-3) / 1e9,
                            "frac": nbytes / (cavg * 1e-3) / 1e9 / HBM_PEAK_GBPS, "launches_timed": len(cold),
                            "buffer_sets": 2}
            out["roofline"] = roof
    return out


def cfg5_config(device):
    """configs[4]: CNF 32768 x 2 (+ logp), dopri5 + adjoint, t 10 -> 0, rtol = atol = 1e-5 — forward / backward ms
    eager and with captured trial steps, and the results vs the reference's (tests/golden/fullsize_cfg5.npz)."""
    import _fullsize as fs
    import torchdiffeq_amd as tda
    z = fs.load("cfg5")
    z0, logp0 = fs.cfg5_problem()
    z0, logp0 = z0.to(device), logp0.to(device)
    t = torch.tensor([10.0, 0.0], device=device)
    idx = torch.from_numpy(z["rows"]).to(device)
    out = {"workload": "BASELINE.json configs[4]: CNF (examples/cnf.py model, closed-form trace), dopri5 + adjoint, "
                       "batch=32768 x dim=2, rtol=atol=1e-5"}
    for name, opts in (("eager", None), ("captured_steps", {"hip_graph": "auto"})):
        cnf = fs.ExampleCNF([z[f"p{i}"] for i in range(6)], trace="closed").to(device)
        cnf.counting = opts is None         # "auto" refuses a func with an evaluation counter (it would stop counting)
        params = list(cnf.parameters())
        best = None
        for rep in range(5):                # (auto: pass 0 eager = first sight, pass 1 captures, passes 2.. replay)
            for p_ in params:
                p_.grad = None
            x = z0.clone().requires_grad_(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            zt, lp = tda.odeint_adjoint(cnf, (x, logp0), t, atol=1e-5, rtol=1e-5, method="dopri5", options=opts)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            loss = lp[-1].mean() - zt[-1].pow(2).sum() / 100
            loss.backward()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if rep >= 2 and (best is None or t2 - t0 < best[0] + best[1]):
                best = (t1 - t0, t2 - t1)
        gp = max(float((p_.grad.cpu() - torch.from_numpy(z[f"grad_p{i}"])).abs().max() /
                       torch.from_numpy(z[f"grad_p{i}"]).abs().max()) for i, p_ in enumerate(params))
        out[name] = {"options": opts, "fwd_ms": 1e3 * best[0], "bwd_ms": 1e3 * best[1],
                     "rel_err_z": fs.sample_rel_err(zt[-1][idx], z["z_end_rows"], z["z_end_absmax"]),
                     "rel_err_logp": fs.sample_rel_err(lp[-1][idx], z["logp_end_rows"], z["logp_end_absmax"]),
                     "rel_err_loss": abs(float(loss.detach()) - float(z["loss"])) / abs(float(z["loss"])),
                     "rel_err_grad_z0": fs.sample_rel_err(x.grad[idx], z["grad_z0_rows"], z["grad_z0_absmax"]),
                     "max_rel_err_param_grads": gp}
    out["reference_1thread_s"] = [float(v) for v in z["wall_s_1thread"]]
    return out


def cfg1_config(device):
    """configs[0]: spiral, rk4, y0 in R^2, 999 steps, fp32 — on the GPU (eager and one captured step replayed) and,
    as BASELINE.json writes it, on the CPU through the package's host path; the reference's trajectory is the
    golden tests/golden/solves.npz."""
    import numpy as np
    import torchdiffeq_amd as tda
    z = np.load(os.path.join(ROOT, "tests", "golden", "solves.npz"))
    ref = torch.from_numpy(z["cfg1_y"])
    out = {"workload": "BASELINE.json configs[0]: spiral ODE, rk4 fixed step, y0 in R^2, batch=1, fp32, 1000 output times"}
    for name, dev_, opts in (("gpu_eager", device, None), ("gpu_captured_step", device, {"hip_graph": True}),
                             ("gpu_auto", device, {"hip_graph": "auto"}), ("cpu_host_path", torch.device("cpu"), None)):
        try:
            A = torch.from_numpy(z["cfg1_A"]).to(dev_)
            y0 = torch.from_numpy(z["cfg1_y0"]).to(dev_)
            t = torch.from_numpy(z["cfg1_t"]).to(dev_)
            f = lambda t_, y_: (y_ ** 3) @ A
            with torch.no_grad():
                tda.odeint(f, y0, t, method="rk4", options=opts)
                if dev_.type == "cuda":
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                y = tda.odeint(f, y0, t, method="rk4", options=opts)
                if dev_.type == "cuda":
                    torch.cuda.synchronize()
                wall = time.perf_counter() - t0
            yc = y.cpu()
            out[name] = {"wall_s": wall, "bit_identical_to_reference": bool(torch.equal(yc, ref)),
                         "rel_err_vs_reference": float((yc - ref).abs().max() / ref.abs().max()),
                         "y_end": yc[-1, 0].tolist()}
        except Exception as exc:
            out[name] = {"error": repr(exc)}
    out["reference_cpu_s"] = 0.134
    return out


def other_configs(device):
    out = {}
    for name, fn in (("cfg4", cfg4_config), ("cfg5", cfg5_config), ("cfg1", cfg1_config)):
        t0 = time.perf_counter()
        try:
            out[name] = fn(device)
        except Exception as exc:
            out[name] = {"error": repr(exc)}
        out[name]["measured_in_s"] = round(time.perf_counter() - t0, 2)
        torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------
# adjoint workload (cfg3)
# ---------------------------------------------------------------------------------------------------
class AllReduceProbe:
    """Counts and times torch.distributed.all_reduce calls (device-synchronised on both sides) while active."""

    def __init__(self):
        self.calls, self.bytes, self.seconds = 0, 0, 0.0
        self._orig = None

    def __enter__(self):
        import torch.distributed as dist
        self._orig = dist.all_reduce

        def probed(tensor, *a, **kw):
            if tensor.is_cuda:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = self._orig(tensor, *a, **kw)
            if tensor.is_cuda:
                torch.cuda.synchronize()
            self.seconds += time.perf_counter() - t0
            self.calls += 1
            self.bytes += tensor.numel() * tensor.element_size()
            return r
        dist.all_reduce = probed
        return self

    def __exit__(self, *exc):
        import torch.distributed as dist
        dist.all_reduce = self._orig


def adjoint_pass(world, rank, device, rows_per_rank, steps, warmup, group_forced=False, graph=False):
    """cfg3 on this rank's `rows_per_rank` rows: K forward + backward passes through odeint_adjoint_sharded.
    graph=True: options={'hip_graph': 'auto'} (forward and, inherited, backward solve as captured trial steps where the
    state is small enough); the field's own Python evaluation counter does not run during replays, so the evaluation
    counts of such a pass are not reported."""
    import _fullsize as fs
    from torchdiffeq_amd import dist as tdist
    field, y0_all = fs.cfg3_problem()
    if rows_per_rank * world <= ADJ_BATCH:
        lo = rank * rows_per_rank
        y0 = y0_all[lo:lo + rows_per_rank].clone()
    else:       # weak scaling: every rank its own 65536 rows (rank 0 = the survey's)
        y0 = y0_all if rank == 0 else torch.randn(rows_per_rank, ADJ_DIM, generator=torch.Generator().manual_seed(rank))
    field = field.to(device)
    y0 = y0.to(device)
    t = torch.tensor([0.0, 1.0], device=device)
    params = list(field.parameters())
    stats = {}
    group = torch.distributed.group.WORLD if (world > 1 or group_forced) else None
    extra = {"options": {"hip_graph": "auto"}} if graph else {}
    if graph:
        # "auto" replays only funcs without per-evaluation side effects (_graph._side_effect_fingerprint): the field's
        # evaluation counter is switched off for this leg
        field.counting = False

    def one():
        for p in params:
            p.grad = None
        x = y0.clone().requires_grad_(True)
        field.nfe = 0
        y = tdist.odeint_adjoint_sharded(field, x, t, group=group, rtol=1e-5, atol=1e-7, method="dopri5", **extra)
        stats["nfe_fwd"], field.nfe = field.nfe, 0
        y[-1].pow(2).sum().backward()
        stats["nfe_bwd"] = field.nfe
    blocks = timed_blocks(one, steps, warmup, world, device, n_blocks=3)
    # launch gaps or kernel floor, per rank: one more pass under the kernel-activity profiler
    breakdown = strong_breakdown(one, 1, block_stats(blocks, steps)["median"], world, rank)
    breakdown["unit_note"] = "per forward + backward pass; func = the MLP, its autograd VJPs and torch glue kernels"
    if graph:
        st = block_stats(blocks, steps)
        return {"rows_per_gpu": rows_per_rank, "ms_per_pass": st["median"], "blocks": st, "options": extra["options"],
                "breakdown": breakdown}
    # one more instrumented pass: forward / backward split and the all-reduce on its own clock
    dist_sync(world)
    with AllReduceProbe() as probe:
        for p in params:
            p.grad = None
        x = y0.clone().requires_grad_(True)
        t0 = time.perf_counter()
        y = tdist.odeint_adjoint_sharded(field, x, t, group=group, rtol=1e-5, atol=1e-7, method="dopri5", **extra)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        y[-1].pow(2).sum().backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    st = block_stats(blocks, steps)
    stages = (stats["nfe_fwd"] - 2) + (stats["nfe_bwd"] - 2)
    grad_norm = float(torch.cat([p.grad.reshape(-1) for p in params]).double().norm())
    return {"rows_per_gpu": rows_per_rank, "ms_per_pass": st["median"], "blocks": st,
            "fwd_ms": 1e3 * max_over_ranks(t1 - t0, world, device),
            "bwd_ms_incl_allreduce": 1e3 * max_over_ranks(t2 - t1, world, device),
            "nfe_fwd": stats["nfe_fwd"], "nfe_bwd": stats["nfe_bwd"], "rk_stages_per_pass": stages,
            "allreduce": {"calls": probe.calls, "bytes": probe.bytes, "ms": 1e3 * probe.seconds,
                          "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
                          "what": "parameter adjoints: the contiguous tail of the flat augmented state, one call "
                                  "(reference: adj_params = aug_state[3:], adjoint.py:150-153)"},
            "param_grad_l2": grad_norm, "breakdown": breakdown}


def adjoint_same_device_reference(device):
    """cfg3 at full size twice on this GPU — HIP kernels vs the reference's own torch expressions (`reference_arithmetic`)
    — with the SAME field arithmetic (hipBLASLt GEMMs, device tanh) on both sides: evaluation counts of the forward and
    backward solve, and how far the gradients are apart.  Answers whether the backward solve's +12 evaluations over the
    reference's CPU run (74 -> 86) come from the solver or from the device's field arithmetic."""
    import contextlib
    import _fullsize as fs
    import torchdiffeq_amd as tda
    field, y0 = fs.cfg3_problem()
    field, y0 = field.to(device), y0.to(device)
    t = torch.tensor([0.0, 1.0], device=device)
    params = list(field.parameters())
    runs = {}
    for name, ctx in (("hip", contextlib.nullcontext()), ("reference_arithmetic", reference_arithmetic())):
        for p in params:
            p.grad = None
        x = y0.clone().requires_grad_(True)
        with ctx:
            field.nfe = 0
            y = tda.odeint_adjoint(field, x, t, rtol=1e-5, atol=1e-7, method="dopri5")
            nfe_fwd, field.nfe = field.nfe, 0
            y[-1].pow(2).sum().backward()
        runs[name] = (nfe_fwd, field.nfe, y[-1].detach(), x.grad.clone(), [p.grad.clone() for p in params])
    a, b = runs["hip"], runs["reference_arithmetic"]
    rel = lambda p, q: float((p - q).abs().max() / q.abs().max())
    return {"what": "cfg3 (odeint_adjoint, MLP 64-256-256-64, 65536 x 64 fp32) on this GPU: HIP kernels vs the reference's own "
                    "torch expressions, same field arithmetic on both sides",
            "nfe_fwd": a[0], "nfe_bwd": a[1], "reference_arithmetic_nfe_fwd": b[0], "reference_arithmetic_nfe_bwd": b[1],
            "reference_on_cpu_nfe": [20, 74],
            "rel_err_y_end": rel(a[2], b[2]), "rel_err_grad_y0": rel(a[3], b[3]),
            "max_rel_err_param_grads": max(rel(p, q) for p, q in zip(a[4], b[4]))}


def run_adjoint(args, rank, world, device):
    strong = args.scaling == "strong"
    rows = ADJ_BATCH // world if strong else ADJ_BATCH
    r = adjoint_pass(world, rank, device, rows, args.steps, args.warmup)
    if rank != 0:
        return None
    # strong: a stage of the global batch is done when every shard's is; weak: ranks' stages add up
    value = r["rk_stages_per_pass"] / (r["ms_per_pass"] * 1e-3) * (1 if strong else world)
    return {
        "metric": "odeint_adjoint RK-stages/sec (dopri5 forward + augmented backward solve, incl. func, its VJPs and "
                  "the parameter-gradient all-reduce)",
        "value": value, "unit": "RK-stages/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["ms_per_pass"], "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[2]: odeint_adjoint, MLP 64-256-256-64 (P=98,880), batch=65536 x "
                               "dim=64 fp32 " + ("in total, rows split over the GPUs" if strong else "per GPU")
                               + ", rtol=1e-5 atol=1e-7, loss sum(y(1)^2)",
                   "global_batch": ADJ_BATCH if strong else ADJ_BATCH * world, "rows_per_gpu": rows,
                   "parallelism": f"batch-sharded x{world}, one all-reduce of the parameter adjoints per backward"},
        "adjoint": r,
    }


# ---------------------------------------------------------------------------------------------------
# launch
# ---------------------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------------
# the contract line (what the driver parses) and the extras file (everything else)
# ---------------------------------------------------------------------------------------------------
CONTRACT_MAX_BYTES = 4096


def _num(x, digits=6):
    """Scalars only: floats rounded to `digits` significant figures, everything that is not a number -> None."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return x
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if x == x and abs(x) != float("inf") else None
    return None


def _short(s, limit=96):
    return s if s is None or len(s) <= limit else s[:limit - 1] + "~"


def contract_line(out, extras_path=None):
    """The ONE line the driver parses: scalars and short tokens only — no prose, no per-kernel tables, no per-rank
    arrays (min / max over ranks as scalars).  Everything else `out` holds goes to the extras file.  Same shape at
    every N and for both workloads; tests/test_bench_line.py pins len(line) < CONTRACT_MAX_BYTES."""
    cfg = out.get("config") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["metric"] = _short(line["metric"], 120)
    line["value"], line["ms_per_step"] = _num(line["value"], 7), _num(line["ms_per_step"], 7)
    line["config"] = {k: (_short(v, 160) if isinstance(v, str) else v) for k, v in cfg.items()
                      if isinstance(v, (str, int, float, bool)) or v is None}
    blocks = (out.get("blocks") or {}).get("ms_per_step") or {}
    if blocks:
        line["block_ms_per_step"] = {k: _num(blocks.get(k)) for k in ("min", "median", "max")}
    per_rank = (out.get("blocks") or {}).get("per_rank_ms_per_step")
    if per_rank:
        line["rank_ms_per_step"] = {"min": _num(min(per_rank)), "max": _num(max(per_rank))}
    rf = out.get("roofline")
    if isinstance(rf, dict):
        cold = rf.get("cold") if isinstance(rf.get("cold"), dict) else {}
        line["roofline"] = {
            "bound": rf.get("bound"), "kernel": _short(rf.get("kernel"), 72), "achieved": _num(rf.get("achieved")),
            "peak": rf.get("peak"), "unit": rf.get("unit"),
            # frac = IN SITU: algorithmic bytes / this kernel's average launch duration inside the timed region (the
            # figure the committed rocprofv3 --stats summary must agree with; the launch's inputs were just written by
            # func, so the 256 MiB Infinity Cache serves part of the reads).  frac_hbm_cold = the same kernel on
            # rotating buffer sets several times that cache: every byte from DRAM.
            "frac": _num(rf.get("frac")), "frac_is": "in_situ", "frac_in_situ": _num(rf.get("frac")),
            "frac_hbm_cold": _num(rf.get("frac_hbm_cold", cold.get("frac"))),
            "algorithmic_bytes_per_launch": rf.get("algorithmic_bytes_per_launch"),
            "avg_launch_ms": _num(rf.get("avg_launch_ms")), "avg_launch_ms_cold": _num(cold.get("avg_launch_ms")),
            "launches_timed": rf.get("launches_timed"),
            # FETCH_SIZE + WRITE_SIZE per launch with the guide's gfx950 corrections: requests on the L2's fabric side,
            # Infinity-Cache hits included (MI355X_MICROARCH.md, HBM section) - an upper bound on DRAM bytes
            "traffic": _num(rf.get("traffic"), 9), "traffic_counts": "l2_fabric_bytes",
            "l2_fabric_bytes": _num(rf.get("traffic"), 9),
            "traffic_source": _short(os.path.basename((rf.get("traffic_source") or "").split(" ")[0]) or None, 48)}
    so = out.get("solver_only")
    if isinstance(so, dict) and "error" not in so:
        line["solver_only"] = {k: _num(so.get(k)) for k in ("stages_per_s", "us_per_step", "GBps_moved",
                                                            "frac_of_hbm_peak_moved") if k in so}
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"),
                                "kind": cb.get("kind"), "sample": _short(cb.get("sample"), 120),
                                "reference_8core_value": _num((cb.get("reference_8core") or {}).get("value"))}
        tc = cb.get("reference_op_sequence_on_torch_cpu")
        if isinstance(tc, dict) and "value" in tc:
            line["cpu_baseline"]["torch_cpu_value"], line["cpu_baseline"]["torch_cpu_cores"] = \
                _num(tc["value"]), tc.get("cores")
    for k in ("rel_err_vs_reference", "rel_err", "nfe", "reference_nfe", "rccl_ranks", "backend",
              "ms_per_pass", "nfe_fwd", "nfe_bwd", "extras_timed_out", "extras_hung_in"):
        if k in out:
            v = out[k]
            line[k] = _num(v) if isinstance(v, float) else (_short(v, 64) if isinstance(v, str) else v)
    adj = out.get("adjoint") if isinstance(out.get("adjoint"), dict) else {}
    for k in ("ms_per_pass", "nfe_fwd", "nfe_bwd", "fwd_ms", "bwd_ms", "rk_stages_per_pass"):
        if k in adj and isinstance(adj[k], (int, float)):
            line[k] = _num(adj[k])
    if "bwd_ms_incl_allreduce" in adj:
        line["bwd_ms"] = _num(adj["bwd_ms_incl_allreduce"])
    # N > 1 linear line: the other regimes the same ranks ran, one scalar pair each (tables in the extras file)
    for regime in ("weak", "strong", "lockstep"):
        r = out.get(regime)
        if isinstance(r, dict) and "value" in r:
            line[regime] = {"value": _num(r.get("value"), 7), "ms_per_step": _num(r.get("ms_per_step"), 7)}
    for mode in ("strong", "weak", "strong_hip_graph_auto"):
        r = adj.get(mode)
        if isinstance(r, dict) and "ms_per_pass" in r:
            line.setdefault("adjoint", {})[mode] = {
                "ms_per_pass": _num(r["ms_per_pass"]), "rk_stages_per_s": _num(r.get("rk_stages_per_s")),
                "nfe_fwd": r.get("nfe_fwd"), "nfe_bwd": r.get("nfe_bwd"),
                "allreduce_calls": (r.get("allreduce") or {}).get("calls"),
                "allreduce_ms": _num((r.get("allreduce") or {}).get("ms"))}
    ar = out.get("allreduce") or adj.get("allreduce")
    if isinstance(ar, dict):
        line["allreduce"] = {k: _num(ar.get(k)) for k in ("calls", "bytes", "ms") if k in ar}
    if out.get("note"):
        line["note"] = _short(out["note"], 120)
    line["extras_file"] = extras_path
    text = json.dumps(line)
    if len(text) >= CONTRACT_MAX_BYTES:      # cannot happen with the fields above; never emit an unparseable line
        for k in ("solver_only", "block_ms_per_step", "rank_ms_per_step", "note", "allreduce", "adjoint", "lockstep"):
            line.pop(k, None)
        text = json.dumps(line)
    assert len(text) < CONTRACT_MAX_BYTES, len(text)
    return text


def write_extras(out):
    """Everything measured (per-kernel breakdowns, the other configs, per-rank tables, definitions in prose) as one
    JSON file: gpurun_out/bench_extras_n{N}.json (TDEQ_BENCH_EXTRAS_DIR overrides the directory).  Returns the path
    relative to the repo root, or None when nothing could be written."""
    d = os.environ.get("TDEQ_BENCH_EXTRAS_DIR") or os.path.join(ROOT, "gpurun_out")
    name = f"bench_extras_n{out.get('n_gpus', 1)}" + \
        ("_adjoint" if str(out.get("metric", "")).startswith("odeint_adjoint") else "") + ".json"
    try:
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, name)
        with open(path, "w") as fh:
            json.dump(out, fh, indent=1, default=repr)
        return os.path.relpath(path, ROOT)
    except Exception:
        return None


def emit(out):
    """Extras to their file (and a digest to stderr); the contract line LAST on stdout, alone."""
    path = write_extras(out)
    sys.stderr.write(f"bench.py: extras -> {path}\n")
    sys.stderr.flush()
    print(contract_line(out, path), flush=True)


class _Watchdog:
    """Fires once after `seconds`: rank 0 prints the JSON line built so far, marked `extras_timed_out` and naming the
    extra measurement that was running (`extras_hung_in`), then the process exits WITHOUT waiting for anything
    (os._exit: a hung collective cannot be joined).  The contract line's own measurement is complete by then, so the
    exit status stays 0; the marker says the line is truncated."""

    def __init__(self, seconds, rank, out, progress):
        import threading
        self._timer = threading.Timer(seconds, self._fire, args=(rank, out, progress, seconds))
        self._timer.daemon = True
        self._timer.start()

    @staticmethod
    def _fire(rank, out, progress, seconds):
        if rank == 0 and out is not None:
            try:
                line = dict(out)
                line["extras_timed_out"] = True
                line["extras_hung_in"] = progress.get("current")
                line["extras_done"] = list(progress.get("done", ()))
                line["extras_timeout_s"] = seconds
                emit(line)
            except Exception:
                pass
        os._exit(0)

    def cancel(self):
        self._timer.cancel()


class _Extras:
    """Names and times the measurements added to the contract line; the watchdog reads `progress`."""

    def __init__(self, rank, out):
        self.rank, self.out = rank, out
        self.progress = {"current": None, "done": []}
        self.seconds = {}

    def run(self, name, fn):
        """fn() -> value for out[name] (rank 0 keeps it); an exception becomes {"error": ...}, never a lost line."""
        self.progress["current"] = name
        t0 = time.perf_counter()
        try:
            val = fn()
        except Exception as exc:
            val = {"error": repr(exc)}
        self.seconds[name] = round(time.perf_counter() - t0, 3)
        self.progress["done"].append(name)
        self.progress["current"] = None
        if self.rank == 0 and self.out is not None and val is not None:
            self.out[name] = val
        return val


def error_line(args, message, **extra):
    """One JSON line saying why no measurement was made (printed by the process that found out)."""
    line = {"metric": "dopri5 RK-stages/sec at batch=65536x dim=128", "value": None, "unit": "RK-stages/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "error": message}
    line.update(extra)
    print(json.dumps(line), flush=True)


def comm_census(rank, world, device):
    """What the collective backend really connects: under nccl (= RCCL) a ones tensor is all-reduced ON THE DEVICE —
    `rccl_ranks` is the number of ranks RCCL summed over — and every rank reports the GPU it sits on.  Returns the
    dict for the JSON line (identical on all ranks) and whether it is consistent with `world`."""
    import torch.distributed as dist
    props = torch.cuda.get_device_properties(device)
    mine = {"rank": rank, "device_index": device.index, "device_name": props.name,
            "device_uuid": str(getattr(props, "uuid", "")), "pid": os.getpid(),
            "visible_devices": torch.cuda.device_count()}
    if world == 1 and not dist.is_initialized():
        return {"backend": None, "rccl_ranks": None, "comm_ranks": 1, "devices": [mine]}, True
    backend = dist.get_backend()
    ones = torch.ones(1, dtype=torch.float32, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(ones)
    if ones.is_cuda:
        torch.cuda.synchronize()
    seen = int(round(float(ones.item())))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    census = {"backend": backend, "rccl_ranks": seen if backend == "nccl" else None, "comm_ranks": seen,
              "devices": gathered}
    ok = seen == world
    if backend == "nccl":       # one rank per GPU: all device identities distinct
        ids = {(d["device_uuid"] or d["device_index"]) for d in gathered}
        idx = {d["device_index"] for d in gathered}
        ok = ok and len(ids) == world and len(idx) == world
    return census, ok


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` as a plain command: start N ranks of this script under torch.distributed.run
    (backend nccl = RCCL, rank r owns GPU r).  More ranks than visible GPUs is REFUSED (one JSON error line, exit
    status 2) unless TDEQ_DIST_BACKEND names a backend explicitly (gloo: ranks share devices — a smoke test of the
    control flow, labelled as such in the line)."""
    env = dict(os.environ)
    visible = torch.cuda.device_count()
    forced = env.get("TDEQ_DIST_BACKEND")
    if args.gpus > visible and not forced:
        error_line(args, f"--gpus {args.gpus} but only {visible} GPU(s) are visible: RCCL needs one GPU per rank. "
                         "Not falling back to another backend silently; set TDEQ_DIST_BACKEND=gloo to smoke-test the "
                         "multi-rank control flow with ranks sharing devices.", visible_devices=visible)
        return 2
    if args.gpus > visible:
        env["TDEQ_BENCH_NOTE"] = f"{args.gpus} ranks on {visible} visible GPU(s): ranks share devices, backend {forced}"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=("linear", "adjoint"), default="linear")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="default: strong when --gpus > 1 (BASELINE.json's metric is quoted at batch 65536 in total: the "
                         "ONE batch split over the GPUs), weak = every GPU its own 65536 rows")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the contract line's own measurement")
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"
    if args.steps is None:
        args.steps = 200 if args.workload == "linear" else 5
    if args.warmup is None:
        args.warmup = 20 if args.workload == "linear" else 2

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    from torchdiffeq_amd import dist as tdist
    # Backend: nccl (= RCCL), rank r owns GPU r.  TDEQ_DIST_BACKEND=gloo lets the N>1 control flow be smoke-tested on
    # a 1-GPU box (ranks share the device; RCCL itself refuses two ranks on one GPU) — only when asked for by name.
    forced = os.environ.get("TDEQ_DIST_BACKEND") or None
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    visible = torch.cuda.device_count()
    if world_env > 1 and world_env > visible and not forced:
        if int(os.environ.get("RANK", "0")) == 0:
            error_line(args, f"WORLD_SIZE={world_env} but only {visible} GPU(s) are visible: RCCL needs one GPU per "
                             "rank; refusing instead of falling back to another backend (TDEQ_DIST_BACKEND=gloo runs "
                             "the control flow with ranks sharing devices).", visible_devices=visible)
        sys.exit(2)
    rank, world, local_rank = tdist.init_from_env(backend=forced)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    local_rank = local_rank % max(visible, 1)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    # what the collective backend really connects — BEFORE anything is timed; an N-rank run whose RCCL communicator
    # does not span N distinct GPUs does not produce a line that looks like a measurement
    census, census_ok = comm_census(rank, world, device)
    if not census_ok:
        if rank == 0:
            error_line(args, f"communicator check failed: all-reduce of ones over backend {census['backend']} gave "
                             f"{census['comm_ranks']} for world size {world}, or ranks share a GPU", comm=census)
        sys.exit(3)

    if args.workload == "adjoint":
        out = run_adjoint(args, rank, world, device)
        if rank == 0:
            out["comm"] = census
            out["rccl_ranks"], out["backend"] = census["rccl_ranks"], census["backend"]
    else:
        out, field, y0 = run_linear(args, rank, world, device)
        if rank == 0:
            out["comm"] = census
            out["rccl_ranks"], out["backend"] = census["rccl_ranks"], census["backend"]
        extras = not args.no_extras
        # The contract line is measured; everything below only adds objects to it.  If an extra hangs (a collective
        # that never completes on some node), every rank's watchdog fires after the same delay: rank 0 prints the
        # line as far as it got — marked, with the name of the measurement that hung — and all ranks leave.
        ex = _Extras(rank, out)
        watchdog = _Watchdog(float(os.environ.get("TDEQ_BENCH_EXTRAS_TIMEOUT", "240")), rank, out, ex.progress)
        if extras and world > 1:
            # the same ranks on the other regime and on the workload that communicates (short runs)
            other = argparse.Namespace(**vars(args))
            other.scaling = "strong" if args.scaling == "weak" else "weak"
            other.steps, other.warmup = min(args.steps, 100), min(args.warmup, 20)

            def other_regime():
                o2, _, _ = run_linear(other, rank, world, device, parity=False)
                return None if o2 is None else {k: o2[k] for k in ("value", "unit", "ms_per_step", "scaling", "config",
                                                                      "blocks")}
            ex.run(other.scaling, other_regime)
            if rank == 0:       # the contract line's own regime under its name too, so both are always there
                out[args.scaling] = {k: out[k] for k in ("value", "unit", "ms_per_step", "scaling", "blocks")}

            def lockstep():
                # lock-step mode (every shard takes the whole-batch step sequence): one all-reduce of 3 doubles per
                # trial step, on the device with RCCL (finalize -> all-reduce -> tdeq_step_controller)
                A_, y0_ = make_problem(device, seed_offset=rank)
                At_ = A_.T.contiguous()
                ls = make_stepper(lambda t, y: y @ At_, y0_, dist_sync=torch.distributed.group.WORLD)
                n_ls = min(args.steps, 100)
                stl = block_stats(time_steps(ls, n_ls, min(args.warmup, 20), world, device, n_blocks=3), n_ls)
                return {"ms_per_step": stl["median"], "blocks": stl, "scaling": "weak",
                        "value": 6e3 / stl["median"] * world, "unit": "RK-stages/s",
                        "lookahead": bool(ls._lookahead),
                        "collective": "all_reduce of the 3 norm words per trial step, "
                                      + ("on the device (RCCL)" if ls._plan_dev is not None else
                                         "through the host (backend without device buffers)")}
            ex.run("lockstep", lockstep)

            adj = {}

            def adjoint_modes():
                for mode, rows in (("strong", ADJ_BATCH // world), ("weak", ADJ_BATCH)):
                    ex.progress["current"] = "adjoint." + mode
                    try:
                        r = adjoint_pass(world, rank, device, rows, 3, 1)
                        r["rk_stages_per_s"] = r["rk_stages_per_pass"] / (r["ms_per_pass"] * 1e-3) * \
                            (1 if mode == "strong" else world)
                        adj[mode] = r
                    except Exception as exc:
                        adj[mode] = {"error": repr(exc)}
                ex.progress["current"] = "adjoint.strong_hip_graph_auto"
                try:        # the strong split once more with captured trial steps (forward and backward solve)
                    adj["strong_hip_graph_auto"] = adjoint_pass(world, rank, device, ADJ_BATCH // world, 3, 2, graph=True)
                except Exception as exc:
                    adj["strong_hip_graph_auto"] = {"error": repr(exc)}
                return adj
            ex.run("adjoint", adjoint_modes)
        if extras and world == 1 and rank == 0:
            def shard_regime():
                r = {"linear": shard_regime_linear(device),
                     "strong_scaling_prediction": strong_scaling_prediction(device, out["ms_per_step"]),
                     "adjoint": adjoint_pass(1, 0, device, ADJ_BATCH // 8, 3, 1),
                     "adjoint_hip_graph_auto": adjoint_pass(1, 0, device, ADJ_BATCH // 8, 3, 2, graph=True)}
                full = out["ms_per_step"]
                la = r["linear"]
                best = min(v["ms_per_step"] for v in la.values() if isinstance(v, dict) and "ms_per_step" in v)
                la["full_size_ms_per_step"] = full
                la["speedup_of_best_over_full_size"] = full / best
                la["note"] = "per-rank work of an 8-GPU strong-scaling run of cfg2; >= 6 would mean the north star's " \
                             "6x at 8 GPUs holds for a fixed global batch"
                return r
            ex.run("shard_regime", shard_regime)
            full = ex.run("adjoint_full", lambda: adjoint_pass(1, 0, device, ADJ_BATCH, 3, 1))

            def adjoint_prediction():
                # one GPU on the shard an N-GPU strong-scaling run of cfg3 gives it (the all-reduce of 0.4 MB excluded)
                pred = {}
                for n_gpus in (2, 4, 8):
                    r = adjoint_pass(1, 0, device, ADJ_BATCH // n_gpus, 3, 1)
                    pred[str(n_gpus)] = {"rows_per_gpu": ADJ_BATCH // n_gpus, "ms_per_pass": r["ms_per_pass"],
                                         "nfe_fwd": r["nfe_fwd"], "nfe_bwd": r["nfe_bwd"],
                                         "predicted_speedup_over_n1": (full["ms_per_pass"] / r["ms_per_pass"])
                                         if isinstance(full, dict) and "ms_per_pass" in full else None,
                                         "func_floor_ms": r["breakdown"].get("func_floor_ms"),
                                         "solver_floor_ms": r["breakdown"].get("solver_floor_ms")}
                return pred
            ex.run("adjoint_strong_scaling_prediction", adjoint_prediction)
            ex.run("configs", lambda: other_configs(device))
            ex.run("low_precision", lambda: low_precision_regime(device))
            ex.run("vector_tolerances", lambda: vector_tolerance_regime(field, y0, device))

            def default_breakdown():
                with torch.no_grad():
                    b = kernel_breakdown(make_stepper(field, y0)._trial_step, 20)
                return {k: b[k] for k in ("solver_kernel_us", "func_kernel_us", "floor_us", "dispatches_per_call")}
            ex.run("breakdown_default_heuristic", default_breakdown)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            def eager():
                with torch.no_grad():
                    return eager_gpu_baseline(field, y0, 0.05)
            ex.run("reference_style_eager_gpu", eager)
            ex.run("same_device_reference", lambda: same_device_reference(field, y0, device))
            ex.run("adjoint_same_device_reference", lambda: adjoint_same_device_reference(device))
            watchdog.cancel()           # the CPU leg and the TunableOp child are bounded by their own clocks
            out["cpu_baseline"] = cpu_baseline()
            if extras and os.environ.get("TDEQ_BENCH_TUNABLEOP", "1") != "0":
                ex.run("func_lever_tunableop", lambda: tunableop_lever(args))
        watchdog.cancel()
        if rank == 0:
            out["extras_s"] = ex.seconds
    if rank == 0:
        if os.environ.get("TDEQ_BENCH_NOTE"):
            out["note"] = os.environ["TDEQ_BENCH_NOTE"]
        emit(out)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
