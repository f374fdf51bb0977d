#!/usr/bin/env python
"""bench.py — BASELINE.json's headline metric on the MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): dopri5, linear field dy/dt = A y, state
65536 x 128 fp32 per GPU, rtol 1e-7 / atol 1e-9 (reference defaults), synthetic seeded data.
One "step" = one dopri5 trial step of the adaptive solver = 6 RK stages: 6 `stage_combine` launches
interleaved with 6 evaluations of the field (a 65536x128x128 GEMM run by PyTorch-ROCm), one fused
`error_norm` launch, one read-back of the error sum and the host step controller.  The state is
resident in HBM before the timed region.  value = RK stages per second over all ranks (weak scaling:
every rank integrates its own 65536-row shard with its own accept/reject loop, no data-path
collective — SURVEY.md §8e).

Extra objects in the JSON line:
  roofline      dominant kernel = stage_combine with 5 stage terms (tableau row 5: read 5 k_j + y0, write
                y_i = 7 words/element = 234.9 MB per launch at this size; row 6 is the same kernel plus the
                fused partial-error store); its launches inside the TIMED region stamp HIP events with the dispatch's own
                begin / end times (tdeq_stage_combine_timed -> hipExtLaunchKernelGGL) on the launch stream.
  solver_only   the step's solver kernels alone, back to back on the last step's stage tensors.
  reference_style_eager_gpu   the reference's eager op sequence restated in stock PyTorch (oracle/eager_torch_port.py)
                timed on the same GPU and state: what a user of the reference gets on an MI355X today.
  cpu_baseline  the CPU oracle (oracle/reference_solver.py + rk_oracle.c, OpenMP on all host cores) on
                a bounded sample of the same workload (rank 0, N=1 only).
  rel_err       max rel-err of a full odeint(t=[0,1]) at this size vs the closed form y0 expm(A)^T.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH, DIM = 65536, 128
RTOL, ATOL = 1e-7, 1e-9
HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def make_problem(device, seed_offset=0):
    """SURVEY.md §8(d) cfg2 synthetic inputs (per-rank seed offset for the weak-scaling shards)."""
    g = torch.Generator().manual_seed(0)
    G = torch.randn(DIM, DIM, generator=g, dtype=torch.float64) / DIM ** 0.5
    A = (0.5 * (G - G.T) - 0.1 * torch.eye(DIM, dtype=torch.float64)).float()
    # rank 0 draws y0 from the same generator right after A (exactly the survey's cfg2 inputs);
    # other ranks draw their own 65536 rows from generator seed = rank.
    gy = g if seed_offset == 0 else torch.Generator().manual_seed(seed_offset)
    y0 = torch.randn(BATCH, DIM, generator=gy, dtype=torch.float64).float()
    return A.to(device), y0.to(device)


class EventTimedKernels:
    """Forwards to HipKernels; while `armed`, the dominant kernel's launches go through tdeq_stage_combine_timed,
    whose dispatch stamps a pair of HIP events with its own begin / end timestamps (hipExtLaunchKernelGGL)."""

    EVERY = 4      # an event-stamped dispatch costs a few microseconds of pipeline: sample every 4th launch

    def __init__(self, inner, dominant_terms, n_events):
        self._seen = 0
        self._inner = inner
        self._nt = dominant_terms
        self.armed = False
        self.events = []
        # events are created (and recorded once: torch creates the hipEvent_t lazily) before the timed region
        self._pool = []
        for _ in range(n_events):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            self._pool.append((e0, e1))

    def __getattr__(self, name):
        return getattr(self._inner, name)

    def stage_combine(self, out, y0, ks, coefs, dt):
        if self.armed and len(ks) == self._nt:
            self._seen += 1
            if self._seen % self.EVERY == 0 and self._pool:
                e0, e1 = self._pool.pop()
                self._inner.stage_combine_timed(out, y0, ks, coefs, dt, e0, e1)
                self.events.append((e0, e1))
                return
        self._inner.stage_combine(out, y0, ks, coefs, dt)


def cpu_baseline(max_seconds=20.0):
    """Oracle (port of the reference algorithm) timed on this host's cores on a bounded sample."""
    import numpy as np
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
    return {"value": 6 * steps / dt, "unit": "RK-stages/s", "cores": cores, "kind": "port",
            "sample": f"{steps} dopri5 trial steps ({6 * steps} RK stages) of the same 65536x128 fp32 workload, "
                      f"oracle/rk_oracle.c with OpenMP on {cores} threads + numpy GEMM, {dt:.1f} s"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from torchdiffeq_amd import _native, dist as tdist
    from torchdiffeq_amd.misc import OdeFunc, StateLayout, rms_norm
    from torchdiffeq_amd.solvers import Dopri5Solver
    import torchdiffeq_amd as tda

    # TDEQ_DIST_BACKEND=gloo lets the N>1 control flow be smoke-tested on a 1-GPU box (ranks share the device;
    # RCCL itself refuses two ranks on one GPU).  Unset, the backend is nccl (= RCCL) and rank r owns GPU r.
    rank, world, local_rank = tdist.init_from_env(backend=os.environ.get("TDEQ_DIST_BACKEND") or None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    A, y0 = make_problem(device, seed_offset=rank)
    At = A.T.contiguous()
    field = lambda t, y: y @ At

    # ---- parity at full size: whole odeint vs the closed form (size-independent property) ----
    with torch.no_grad():
        t_wall = time.perf_counter()
        y_end = tda.odeint(field, y0, torch.tensor([0.0, 1.0], device=device), rtol=RTOL, atol=ATOL, method="dopri5")[-1]
        torch.cuda.synchronize()
        odeint_wall = time.perf_counter() - t_wall
        exact = y0.double() @ torch.linalg.matrix_exp(A.double()).T
        rel_err = float((y_end.double() - exact).abs().max() / exact.abs().max())

    # ---- timed region: K trial steps of the adaptive solver ----
    layout = StateLayout([y0.shape], False)
    func = OdeFunc(field, layout, 1.0, y0.dtype, device)
    solver = Dopri5Solver(func=func, y0=y0.reshape(-1), rtol=RTOL, atol=ATOL, norm=rms_norm)
    timed = EventTimedKernels(solver.kernels, dominant_terms=5, n_events=args.steps // EventTimedKernels.EVERY + 1)
    solver.kernels = timed
    solver.ops.k = timed        # the elementwise kernels are issued through solver.ops
    with torch.no_grad():
        solver._before_integrate([0.0])
        # the timed trial steps are steps in the middle of a long solve (no output time ahead), which is where the
        # solver's look-ahead first stage applies (solvers.py: only the last steps before the final output time of
        # an `integrate` call are driven without it)
        solver._t_end = float("inf")
        for _ in range(args.warmup):
            solver._adaptive_step()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        timed.armed = True
        t0 = time.perf_counter()
        for _ in range(args.steps):
            solver._adaptive_step()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        elapsed = time.perf_counter() - t0
        timed.armed = False
    if world > 1:
        el = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(el.item())

    # ---- solver-only rate: the 6 stage_combine launches + error_norm (+ finalize) of one
    # dopri5 step on the k tensors of the last timed step, no func, HIP events around REPS back-to-back passes.
    solver_only = None
    try:
        rec = solver._dense
        kern = solver.kernels._inner
        ks, y0s = rec.k, rec.y0
        outs = [torch.empty_like(y0s) for _ in range(2)]
        REPS = 30

        fuse = solver._fuse
        epart = torch.empty_like(y0s)
        last = len(solver._beta) - 1

        la = bool(solver._lookahead and fuse is not None)
        tnext = torch.empty(len(solver._beta), dtype=y0s.dtype, device=device)

        def one_pass():
            # exactly the solver's launch sequence for one trial step, minus func (and, without look-ahead, the
            # stage-time fill)
            for i, row in enumerate(solver._beta):
                if i == 0 and la:
                    kern.stage_combine_sel(outs[0], rec.y1, ks[-1], y0s, ks[0], row.coef[0], solver.plan)
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
        # SURVEY.md §8(d) counts 32 + 8 = 40 words per element for a dopri5 step; the end-of-step fusion moves
        # 37 (3+4+5+6+7+8 for the six combines, 4 for the norm) — both rates are reported.
        moved = (37 if fuse is not None else 40) * BATCH * DIM * 4
        survey = 40 * BATCH * DIM * 4
        solver_only = {"stages_per_s": 6 / t_step, "us_per_step": 1e6 * t_step,
                       "bytes_moved_per_step": moved, "GBps_moved": moved / t_step / 1e9,
                       "frac_of_hbm_peak_moved": moved / t_step / 1e9 / HBM_PEAK_GBPS,
                       "survey_algorithmic_bytes_per_step": survey, "GBps_survey_bytes": survey / t_step / 1e9,
                       "note": "the solver's own launch sequence for one dopri5 trial step (stage_combine_sel + 4 "
                               "stage_combine + stage_combine_err + error_norm_partial + controller finalize) back "
                               "to back, no func; the 7 k tensors (235 MB) + y0/y1 fit the 256 MiB Infinity Cache "
                               "only partly"}
    except Exception as exc:      # never let the extra figure break the contract line
        solver_only = {"error": repr(exc)}

    n = BATCH * DIM
    # dispatch begin -> end of each timed launch (the same interval rocprofv3's kernel trace reports)
    kernel_ms = [a.elapsed_time(b) for a, b in timed.events]
    avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
    bytes_per_launch = 7 * n * 4
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if kernel_ms else None
    # HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/profile_gpu.sh ->
    # profiles/<tag>_pmc_hbm.json; counters cannot be collected from inside this process).
    traffic = None
    import glob
    for pmc_path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json")), reverse=True):
        try:
            kernels = json.load(open(pmc_path))["kernels"]
            hit = [v for k, v in kernels.items() if k.startswith("tdeq::stage_combine_kernel<float, 5,")]
            if hit:
                traffic = hit[0]["hbm_bytes_per_launch"]
                break
        except Exception:
            continue

    if rank == 0:
        out = {
            "metric": "dopri5 RK-stages/sec at batch=65536x dim=128 (end-to-end adaptive trial steps incl. func, "
                      "error norm, read-back and host controller)",
            "value": 6 * args.steps * world / elapsed,
            "unit": "RK-stages/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: dopri5 adaptive, linear ODE dy/dt=Ay, batch=65536 x "
                                   "dim=128 fp32 per GPU, rtol=1e-7 atol=1e-9",
                       "global_batch": BATCH * world, "dim": DIM, "parallelism": f"batch-sharded x{world}",
                       "accepted": solver.n_accepted, "rejected": solver.n_rejected,
                       "lookahead": bool(solver._lookahead)},
            "rel_err": rel_err,
            "rel_err_definition": "max|y - y_exact| / max|y_exact| of odeint(t=[0,1]) at full size vs y0 @ expm(A)^T "
                                  "(the reference's own fp32 result scores 2.2-2.6e-6 on this, SURVEY.md §7)",
            "odeint_t01_wall_s": odeint_wall,
            "roofline": {"bound": "hbm", "kernel": "stage_combine_kernel<float, 5, 1, true>",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms,
                         "timing": "HIP events stamped by the dispatch itself (hipExtLaunchKernelGGL start/stop) on "
                                   "the launch stream, every 4th launch of this kernel in the timed region",
                         "launches_timed": len(kernel_ms), "traffic": traffic},
            "solver_only": solver_only,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                with torch.no_grad():
                    out["reference_style_eager_gpu"] = eager_gpu_baseline(field, y0, 0.05)
            except Exception as exc:      # an extra figure, never allowed to break the contract line
                out["reference_style_eager_gpu"] = {"error": repr(exc)}
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
