"""bench_core.py — what bench.py times: the synthetic problems of SURVEY.md §8(d), the timed-block protocol (barrier +
synchronize on both sides, max over ranks), the trial-step stepper, the live roofline of the dominant launch (dispatch-stamped
HIP events in the timed region, the same kernel HBM-cold on rotating buffers, PMC traffic from the committed summaries), the
kernel-floor breakdown, the two workloads (`run_linear` = BASELINE.json configs[1], `run_adjoint` = configs[2]) and the
communicator census.  `bench.py` holds the contract line and `main()`, `bench_extras.py` everything that only feeds the
extras file."""
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
                # the instantiation this launch resolves to (csrc/tdeq_abi.hip launch_combine_multi: default cache policy,
                # output count / no prefix / dt folded by the host / one pass as compile-time constants) — the name the
                # rocprofv3 summary under profiles/ lists
                shape = f", 0, {len(outs)}, 0, false, true" if len(outs) <= 2 else ""
                self.kernel = f"stage_combine_multi_kernel<float, {len(ks)}, true{shape}> ({len(outs)} outputs)"
                return self._inner.stage_combine_multi(outs, rows, y0, acc_in, ks, dt, events=ev)
        return self._inner.stage_combine_multi(outs, rows, y0, acc_in, ks, dt)


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
            "kernel": "stage_combine_multi_kernel<float, 4, true, 0, 2, 0, false, true> (2 outputs)" if carried
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
