"""bench_extras.py — measurements that only feed the extras file (`gpurun_out/bench_extras_n{N}.json`), never the contract
line: the CPU baseline legs, the reference's eager op sequence and the reference's arithmetic on the same GPU, the 1/8-shard
regime and strong-scaling predictions, the other BASELINE.json configurations (cfg4, cfg5, cfg1), and the r05 regimes
(reduced-precision states, per-element tolerances, the TunableOp lever on `func`)."""
from __future__ import annotations

import json
import os
import statistics
import subprocess
import sys
import time

import torch

from bench_core import (ADJ_BATCH, ADJ_DIM, ATOL, BATCH, DIM, HBM_PEAK_GBPS, ROOT, RTOL, adjoint_pass,  # noqa: F401
                        kernel_breakdown, make_problem, make_stepper, strong_breakdown, time_steps, block_stats)


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


def shard_regime_linear(device, steps=100, warmup=20):
    """One GPU on the 8192 x 128 shard (1/8 of cfg2): ms per trial step on the three step paths."""
    A, y0 = make_problem(device, rows=slice(0, BATCH // 8))
    At = A.T.contiguous()
    field = lambda t, y: y @ At
    out = {"state": f"{BATCH // 8} x {DIM} fp32 (1/8 of cfg2)", "steps_per_block": steps}
    # "default": no option, no environment variable — what a drop-in user gets (r06: hip_graph='auto' built in);
    # "lookahead" / "host_driven": the eager loops, asked for explicitly (hip_graph=False)
    for name, kw in (("host_driven", dict(lookahead=False)), ("lookahead", dict(lookahead=True)),
                     ("hip_graph", dict(hip_graph=True)), ("auto", dict(hip_graph="auto")), ("default", dict(hip_graph=None))):
        try:
            if name == "default":
                field = lambda t, y: y @ At          # a func object "auto" has not seen: the whole default experience
            solver = make_stepper(field, y0, **kw)
            # ("auto" / "default": first sight of this func -> eager until solvers._AUTO_CAPTURE_AFTER_STEPS trial steps, then
            #  captured)
            blocks = time_steps(solver, steps, warmup if name not in ("auto", "default") else warmup + 110, 1, device, n_blocks=3)
            st = block_stats(blocks, steps)
            out[name] = {"ms_per_step": st["median"], "min": st["min"], "max": st["max"],
                         "stages_per_s_of_the_shard": 6e3 / st["median"]}
            if name in ("auto", "default"):
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


def lowp_steps(dtype, backend, steps, warmup, device, hip_graph=False):
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
                solver = make_stepper(lambda t, y: y @ At, y0, rtol=1e-2, atol=1e-3, hip_graph=hip_graph)
                for _ in range(warmup):
                    solver._trial_step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    solver._trial_step()
                torch.cuda.synchronize()
                blocks.append((time.perf_counter() - t0) / steps)
            breakdown = None
            if backend == "hip" and not hip_graph:
                b = kernel_breakdown(solver._trial_step, 10)
                breakdown = {k: b[k] for k in ("solver_kernel_us", "func_kernel_us", "floor_us", "dispatches_per_call",
                                               "top_kernels")}
            if solver._g is not None:
                torch.cuda.synchronize()
                solver._g.release()
    finally:
        _native.get_kernels = orig
    ms = 1e3 * statistics.median(blocks)
    return {"backend": solver.kernels.name, "lookahead": bool(solver._lookahead), "hip_graph": bool(solver.hip_graph),
            "ms_per_step": ms,
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


def lowp_adjoint_pass(backend, device, passes=3):
    """cfg3's shape with a bf16 STATE and bf16 parameters (MLP 64-256-256-64, 65536 x 64, loss sum(y(1)^2), rtol 1e-2 / atol
    1e-3): one forward + backward pass of odeint_adjoint on the 16-bit HIP kernels (device controller + look-ahead in both
    solves) or on the torch-op path forced onto the same device."""
    import torchdiffeq_amd as tda
    from torchdiffeq_amd import _fallback, _native
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(ADJ_DIM, 256), torch.nn.Tanh(), torch.nn.Linear(256, 256), torch.nn.Tanh(),
                              torch.nn.Linear(256, ADJ_DIM)).to(torch.bfloat16).to(device)

    class Field(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net, self.nfe = net, 0

        def forward(self, t, y):
            self.nfe += 1
            return self.net(y)
    field = Field()
    y0 = torch.randn(ADJ_BATCH, ADJ_DIM, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).to(device)
    t = torch.tensor([0.0, 1.0], device=device)
    orig = _native.get_kernels
    if backend == "torch-op":
        low = _fallback.LowPrecisionHostKernels()
        _native.get_kernels = lambda dev_, dt_=None: low if dt_ in (torch.bfloat16, torch.float16) else orig(dev_, dt_)
    try:
        times = []
        for _ in range(passes + 1):
            field.zero_grad()
            x = y0.clone().requires_grad_(True)
            field.nfe = 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y = tda.odeint_adjoint(field, x, t, method="dopri5", rtol=1e-2, atol=1e-3)
            n_fwd, field.nfe = field.nfe, 0
            y[-1].float().pow(2).sum().backward()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    finally:
        _native.get_kernels = orig
    return {"ms_per_pass": 1e3 * statistics.median(times[1:]), "nfe_fwd": n_fwd, "nfe_bwd": field.nfe,
            "grad_finite": bool(torch.isfinite(x.grad.float()).all())}


def low_precision_regime(device):
    res = {"workload": "dopri5 trial steps, dy/dt = A y (rotation), 65536 x 128, rtol 1e-2 atol 1e-3, state in bf16 / fp16"}
    for name, dtype in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        entry = {"stage_combine": lowp_combine_rate(dtype, device)}
        try:
            hip = lowp_steps(dtype, "hip", 40, 5, device)
            ref = lowp_steps(dtype, "torch-op", 10, 2, device)
            # captured trial steps (options={'hip_graph': True}): at this size the eager loop is HOST-bound — six func
            # dispatches + eight ctypes launches of Python per step — so one replay per step reaches the kernel floor
            cap = lowp_steps(dtype, "hip", 40, 8, device, hip_graph=True)
            entry.update({"hip_kernels": hip, "hip_kernels_captured": cap, "torch_op_host_path": ref,
                          "speedup": ref["ms_per_step"] / hip["ms_per_step"],
                          "speedup_captured": ref["ms_per_step"] / cap["ms_per_step"]})
        except AssertionError as exc:
            # float16: the initial-step heuristic underflows the type's range in the reference as well ("underflow in dt
            # 0.0", tests/test_brow_golden.py) — adaptive solves of fp16 states do not start; fixed grids do
            entry["adaptive_steps"] = {"error": str(exc)}
        res[name] = entry
    try:
        hip, ref = lowp_adjoint_pass("hip", device), lowp_adjoint_pass("torch-op", device, passes=2)
        res["bf16"]["adjoint_cfg3_shape"] = {"hip_kernels": hip, "torch_op_host_path": ref,
                                             "speedup": ref["ms_per_pass"] / hip["ms_per_pass"]}
    except Exception as exc:
        res["bf16"]["adjoint_cfg3_shape"] = {"error": repr(exc)}
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
        if fused is False:          # the r04 route: raw error materialised, scaling + norm as fp64 torch ops, host-driven steps
            solver._vec_fused, solver._vec_ctrl, solver._lookahead = None, False, False
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
    out["note"] = "fused = tdeq_error_norm_vec_ctrl (norm + device controller, look-ahead kept); torch_ops = the r04 route"
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
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(args.steps),
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
    # "default": no option (r06: = 'auto' built in) — must equal "captured_steps"; "eager": asked for with hip_graph=False
    for name, opts in (("eager", {"hip_graph": False}), ("captured_steps", {"hip_graph": "auto"}), ("default", None)):
        cnf = fs.ExampleCNF([z[f"p{i}"] for i in range(6)], trace="closed").to(device)
        cnf.counting = name == "eager"      # "auto" refuses a func with an evaluation counter (it would stop counting)
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
    for name, dev_, opts in (("gpu_eager", device, {"hip_graph": False}), ("gpu_captured_step", device, {"hip_graph": True}),
                             ("gpu_auto", device, {"hip_graph": "auto"}), ("gpu_default", device, None),
                             ("cpu_host_path", torch.device("cpu"), None)):
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
