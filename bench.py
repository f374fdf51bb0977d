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

Output: ONE stdout line — the contract line, a whitelist of scalars < 4 KB at every N (`contract_line`): metric, value,
unit, n_gpus, steps, warmup, ms_per_step, scaling, dtype, data, config{workload, ...}, roofline{bound, kernel, achieved,
peak, frac (= in situ), frac_hbm_cold, algorithmic_bytes_per_launch, avg_launch_ms, avg_launch_ms_cold, traffic (=
l2_fabric_bytes per launch from the committed PMC summary)}, cpu_baseline{value, cores, kind, sample,
reference_8core_value}, rel_err_vs_reference, nfe / reference_nfe, rccl_ranks / backend and — N > 1 — one scalar pair per
regime (weak, strong, lockstep, adjoint.{strong, weak}).  EVERYTHING ELSE goes to `gpurun_out/bench_extras_n{N}.json`
(TDEQ_BENCH_EXTRAS_DIR overrides the directory; the line names the file):
  roofline      the launch the north star names = the step's 7-words-per-element stage combine (234.9 MB; the 8-word
                stage_combine_err launch is marginally heavier by total time; with carried partial sums row 4 +
                the prefix of row 5: stage_combine_multi_kernel<float, 4>, 2 outputs).  `frac` = IN SITU: its launches inside
                the timed region, each stamped by the dispatch itself (hipExtLaunchKernelGGL start / stop events) — the stage
                tensors were written by `func` just before, so part of the reads is served by the 256 MiB Infinity Cache;
                `cold` = the same kernel on 4 rotating buffer sets (940 MB): every byte from DRAM.
  solver_only   the step's solver kernels alone, back to back on the last step's stage tensors.
  shard_regime, adjoint_full, adjoint_strong_scaling_prediction   (N = 1) what one GPU does on the shards an N-GPU
                strong-scaling run gives it, with the kernel-floor breakdown.
  configs       (N = 1) the other BASELINE.json configurations, bounded: cfg4 (dopri8 fp64), cfg5 (CNF + adjoint), cfg1.
  low_precision, vector_tolerances, func_lever_tunableop, breakdown_default_heuristic   (N = 1, r05) bf16 / fp16 states on
                the HIP kernels vs the torch-op path; per-element tolerances fused vs the r04 route; the headline step with
                PyTorch's TunableOp choosing func's GEMMs (a child process; never the contract value).
  weak, strong, lockstep, adjoint   (N > 1) the same ranks on both scaling regimes of cfg2, in lock-step mode and on cfg3, with
                per-rank breakdowns; a watchdog (TDEQ_BENCH_EXTRAS_TIMEOUT, 240 s) prints the line marked
                `extras_timed_out` + `extras_hung_in` if one of them hangs.
  comm          what the collective backend connected (see above).
  reference_style_eager_gpu, same_device_reference, adjoint_same_device_reference   the reference's eager op sequence and the
                reference's arithmetic on the same GPU.
  cpu_baseline  the CPU oracle (a port: the reference itself cannot travel to the GPU box) on a bounded sample, on rank 0 at
                N = 1, with BASELINE.md's figure for the real reference on 8 cores beside it.
Code: bench_core.py (problems, timing protocol, workloads, live roofline), bench_extras.py (extras), this file (contract).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics  # noqa: F401
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))      # bench_core / bench_extras live next to this file

# r05: the file is split — bench_core.py (problems, timing protocol, the two workloads, live roofline), bench_extras.py
# (everything that only feeds the extras file); this file keeps the contract: the ONE stdout line and main().  Every name is
# re-exported, so `import bench; bench.make_problem(...)` (tools/, tests/) keeps working.
from bench_core import *  # noqa: F401,F403
from bench_core import (ADJ_BATCH, ATOL, BATCH, DIM, HBM_PEAK_GBPS, N_BLOCKS, ROOT, RTOL, adjoint_pass, block_stats,  # noqa: F401
                        comm_census, kernel_breakdown, make_problem, make_stepper, run_adjoint, run_linear, time_steps)
from bench_extras import *  # noqa: F401,F403
from bench_extras import (adjoint_same_device_reference, cpu_baseline, eager_gpu_baseline, low_precision_regime,  # noqa: F401
                          other_configs, same_device_reference, shard_regime_linear, strong_scaling_prediction,
                          tunableop_lever, vector_tolerance_regime)


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
            "bound": rf.get("bound"), "kernel": _short(rf.get("kernel"), 112), "achieved": _num(rf.get("achieved")),
            # (r06) which launch this is: the one BASELINE.json's north_star names (its target is quoted on the dopri5 stage
            # combine) — by total time the step's 8-word stage_combine_err launch is marginally heavier (profiles/ CSV)
            "kernel_role": "north_star's stage-combine launch, 7 words/element",
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
