"""Per-kernel bandwidth microbenchmark on the MI355X (SURVEY.md §8d "kernel microbench").

Times each hot-path kernel at a BASELINE config's state size with HIP events on the launch stream,
rotating through several buffer sets so the 256 MiB Infinity Cache cannot serve re-reads, and prints
algorithmic GB/s and the fraction of the 8 TB/s HBM peak.

  python tools/kernel_bench.py [--batch 65536 --dim 128 --dtype f32 --method dopri5 --iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from torchdiffeq_amd import _native  # noqa: E402
from torchdiffeq_amd.tableaus import DOPRI5, DOPRI8, SparseRow  # noqa: E402

HBM_PEAK = 8.0e12


def timed(fn, iters, warmup=3):
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    start = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    stop = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        start[i].record()
        fn(i)
        stop[i].record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in zip(start, stop))
    return ms[len(ms) // 2] * 1e-3, ms[0] * 1e-3


def adams_main(args, dtype, w):
    """tdeq_adams_predict at the method's top order (11 history tensors) and tdeq_adams_correct, cold (rotating sets)."""
    from torchdiffeq_amd.tableaus import adams_coefficients
    n = args.batch * args.dim
    dev = torch.device("cuda:0")
    kern = _native.get_kernels(dev)
    g = torch.Generator(device="cuda").manual_seed(1)
    order = 11
    sets = []
    for _ in range(args.sets):
        hist = [torch.randn(n, generator=g, device=dev, dtype=dtype) for _ in range(order)]
        y0 = torch.randn(n, generator=g, device=dev, dtype=dtype)
        outs = [torch.empty(n, device=dev, dtype=dtype) for _ in range(4)]
        sets.append((hist, y0, outs))
    plan = kern.make_plan([(0, n, 1e-7, 1e-9)], n, args.chunk or _native.pick_chunk(n), dev)
    bash, _ = adams_coefficients(order)
    _, moulton = adams_coefficients(order + 1)
    cb = [0.01 * b for b in bash]
    results = []

    def report(name, words, t_med, t_min):
        by = words * n * w
        results.append({"kernel": name, "words_per_elem": words, "bytes": by, "t_med_us": t_med * 1e6,
                        "t_min_us": t_min * 1e6, "GBps_med": by / t_med / 1e9, "frac_of_8TBps": by / t_med / HBM_PEAK})
        print(f"{name:34s} {words:3d} w/elem  {t_med*1e6:9.1f} us (min {t_min*1e6:8.1f})  "
              f"{by/t_med/1e9:8.1f} GB/s  {100*by/t_med/HBM_PEAK:5.1f}% of 8 TB/s", flush=True)

    def fn_explicit(it):
        hist, y0, outs = sets[it % args.sets]
        kern.adams_predict(outs[0], y0, hist, cb)
    report("adams_predict explicit nt=11", order + 2, *timed(fn_explicit, args.iters))

    def fn_implicit(it):
        hist, y0, outs = sets[it % args.sets]
        kern.adams_predict(outs[0], y0, hist, cb, list(moulton[1:]), 0.01, dy_out=outs[1], delta_out=outs[2])
    report("adams_predict implicit nt=11", order + 4, *timed(fn_implicit, args.iters))

    def fn_correct(it):
        hist, y0, outs = sets[it % args.sets]
        kern.adams_correct(plan, outs[3], outs[1], y_out=outs[0], f=hist[0], delta=outs[2], y0=y0, c=0.003)
    report("adams_correct (+census+finalize)", 6, *timed(fn_correct, args.iters))
    print(json.dumps({"n": n, "dtype": args.dtype, "method": "adams", "results": results}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--method", default="dopri5")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sets", type=int, default=3, help="rotating buffer sets")
    ap.add_argument("--chunk", type=int, default=0)
    args = ap.parse_args()
    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    w = 4 if dtype == torch.float32 else 8
    if args.method == "adams":
        return adams_main(args, dtype, w)
    tab = DOPRI5 if args.method == "dopri5" else DOPRI8
    n = args.batch * args.dim
    dev = torch.device("cuda:0")
    kern = _native.get_kernels(dev)
    S = tab.n_stages
    g = torch.Generator(device="cuda").manual_seed(1)
    sets = []
    for _ in range(args.sets):
        ks = [torch.randn(n, generator=g, device=dev, dtype=dtype) for _ in range(S + 1)]
        y0 = torch.randn(n, generator=g, device=dev, dtype=dtype)
        y1 = torch.randn(n, generator=g, device=dev, dtype=dtype)
        out = torch.empty(n, device=dev, dtype=dtype)
        sets.append((ks, y0, y1, out))
    chunk = args.chunk or _native.pick_chunk(n)
    plan = kern.make_plan([(0, n, 1e-7, 1e-9)], n, chunk, dev)
    results = []

    def report(name, words, t_med, t_min):
        by = words * n * w
        results.append({"kernel": name, "words_per_elem": words, "bytes": by, "t_med_us": t_med * 1e6,
                        "t_min_us": t_min * 1e6, "GBps_med": by / t_med / 1e9,
                        "frac_of_8TBps": by / t_med / HBM_PEAK})
        print(f"{name:28s} {words:3d} w/elem  {t_med*1e6:9.1f} us (min {t_min*1e6:8.1f})  "
              f"{by/t_med/1e9:8.1f} GB/s  {100*by/t_med/HBM_PEAK:5.1f}% of 8 TB/s", flush=True)

    total_t, total_words = 0.0, 0
    for i, row in enumerate(tab.beta_rows()):
        def fn(it, row=row):
            ks, y0, y1, out = sets[it % args.sets]
            kern.stage_combine(out, y0, [ks[j] for j in row.idx], row.coef, 0.1)
        t_med, t_min = timed(fn, args.iters)
        report(f"stage_combine row{i+1} nt={len(row.idx)}", len(row.idx) + 2, t_med, t_min)
        total_t += t_med
        total_words += len(row.idx) + 2
    err = SparseRow.from_dense(tab.c_error)

    def fn_err(it):
        ks, y0, y1, out = sets[it % args.sets]
        kern.error_norm(plan, y0, y1, [ks[j] for j in err.idx], err.coef, 0.1)
    t_med, t_min = timed(fn_err, args.iters)
    report(f"error_norm nt={len(err.idx)} chunk={chunk}", len(err.idx) + 2, t_med, t_min)
    total_t += t_med
    total_words += len(err.idx) + 2
    mid = SparseRow.from_dense(tab.c_mid)

    def fn_dense(it):
        ks, y0, y1, out = sets[it % args.sets]
        kern.dense_eval(out, y0, y1, ks[0], ks[-1], [ks[j] for j in mid.idx], mid.coef, 0.1, 0.5)
    t_med, t_min = timed(fn_dense, args.iters)
    report(f"dense_eval nt={len(mid.idx)}", len(mid.idx) + 3, t_med, t_min)

    def fn_copy(it):
        ks, y0, y1, out = sets[it % args.sets]
        out.copy_(y0)
    t_med, t_min = timed(fn_copy, args.iters)
    report("torch copy_ (ref. 2 w/elem)", 2, t_med, t_min)
    by = total_words * n * w
    print(f"solver-only step ({S} combines + error_norm): {total_t*1e6:.1f} us, {by/total_t/1e9:.1f} GB/s, "
          f"{100*by/total_t/HBM_PEAK:.1f}% of 8 TB/s -> {S/total_t:.0f} RK-stages/s")
    print(json.dumps({"n": n, "dtype": args.dtype, "method": args.method, "results": results,
                      "solver_only_stages_per_s": S / total_t}))


if __name__ == "__main__":
    main()
