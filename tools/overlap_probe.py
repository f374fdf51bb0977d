"""Feasibility probe for docs/LAB_NOTEBOOK.md §12's GEMM / combine overlap, run on the MI355X through gpurun.

The dopri5 trial step of cfg2 is serial by data dependence: combine_i -> func_i (a 65536x128x128 fp32 GEMM) -> ...;
the combines are HBM-bound, the GEMM is the compute-heavier block.  If the batch is split in row blocks, the chain of
block A can run beside the chain of block B (different kernels of the two chains in flight at the same time).  This
probe captures ONE trial step's launch sequence (6 stage combines with the shipped kernels, 6 GEMMs, the fused error
norm) into a hipGraph in several arrangements and times replays:

    serial        the whole batch, one chain (what the eager look-ahead loop issues)
    split2        two half-batch chains forked / joined inside the graph, started together
    split2_skew   the same, chain B released only after chain A's first combine (so A's GEMM meets B's combine)
    split4        four quarter-batch chains
    seq2/4/8      the row blocks one after the other (temporal blocking: does a block's working set in the Infinity
                  Cache make its kernels faster than the fixed costs of smaller kernels make them slower?)

All arrangements compute the same values (the split ones on row blocks).  Prints one JSON object."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torchdiffeq_amd import _native  # noqa: E402
from torchdiffeq_amd.tableaus import DOPRI5, SparseRow  # noqa: E402

dev = torch.device("cuda:0")
B, D = 65536, 128
kern = _native.get_kernels(dev)
g = torch.Generator().manual_seed(0)
G = torch.randn(D, D, generator=g, dtype=torch.float64) / D ** 0.5
At = (0.5 * (G - G.T) - 0.1 * torch.eye(D, dtype=torch.float64)).float().T.contiguous().to(dev)
y0_full = torch.randn(B, D, generator=g, dtype=torch.float64).float().to(dev)
beta = DOPRI5.beta_rows()
c_err = SparseRow.from_dense(DOPRI5.c_error)
DT = 0.1


def chain(y0, plan, epart):
    """One trial step's launches on the row block y0 [rows, D]: returns (y1, k list)."""
    flat = y0.reshape(-1)
    k = [(y0 @ At).reshape(-1)]
    yi = None
    for i, row in enumerate(beta):
        yi = torch.empty_like(flat)
        ks = [k[j] for j in row.idx]
        if i == len(beta) - 1:
            kern.stage_combine_err(yi, epart, flat, ks, row.coef, c_err.coef[:len(row.idx)], DT)
        else:
            kern.stage_combine(yi, flat, ks, row.coef, DT)
        k.append((yi.view_as(y0) @ At).reshape(-1))
    kern.error_norm_partial(plan, epart, flat, yi, [k[j] for j in c_err.idx[len(beta[-1].idx):]],
                            c_err.coef[len(beta[-1].idx):], DT)
    return yi, k


def make_parts(n_parts):
    rows = B // n_parts
    parts = []
    for p in range(n_parts):
        y = y0_full[p * rows:(p + 1) * rows].contiguous()
        n = y.numel()
        plan = kern.make_plan([(0, n, 1e-7, 1e-9)], n, _native.pick_chunk(n), dev)
        parts.append((y, plan, torch.empty(n, device=dev)))
    return parts


def capture(n_parts, skew):
    parts = make_parts(n_parts)
    main = torch.cuda.Stream(dev)
    sides = [torch.cuda.Stream(dev) for _ in range(n_parts)]
    keep = []

    def body():
        if n_parts == 1:
            keep.append(chain(*parts[0]))
            return
        if skew == "seq":
            # temporal blocking: the row blocks one AFTER the other on one stream — each block's working set
            # (9 state-sized tensors / n_parts) then fits the 256 MiB Infinity Cache
            for part in parts:
                keep.append(chain(*part))
            return
        cur = torch.cuda.current_stream()
        prev_first = None
        for s, part in zip(sides, parts):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                if skew and prev_first is not None:
                    s.wait_event(prev_first)
                y0, plan, epart = part
                flat = y0.reshape(-1)
                # first GEMM + combine, then an event the next chain may wait for
                k0 = (y0 @ At).reshape(-1)
                ev = torch.cuda.Event()
                ev.record(s)
                prev_first = ev
                k = [k0]
                yi = None
                for i, row in enumerate(beta):
                    yi = torch.empty_like(flat)
                    ks = [k[j] for j in row.idx]
                    if i == len(beta) - 1:
                        kern.stage_combine_err(yi, epart, flat, ks, row.coef, c_err.coef[:len(row.idx)], DT)
                    else:
                        kern.stage_combine(yi, flat, ks, row.coef, DT)
                    k.append((yi.view_as(y0) @ At).reshape(-1))
                kern.error_norm_partial(plan, epart, flat, yi, [k[j] for j in c_err.idx[len(beta[-1].idx):]],
                                        c_err.coef[len(beta[-1].idx):], DT)
                keep.append((yi, k))
        for s in sides:
            cur.wait_stream(s)

    with torch.cuda.stream(main):
        body()              # warm-up (allocator, hipBLASLt heuristics)
        keep.clear()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        graph.capture_begin()
        body()
        graph.capture_end()
    torch.cuda.synchronize()
    for _, plan, _ in parts:
        plan.expect = ()
    return graph, keep, parts


def time_graph(graph, reps=200):
    for _ in range(20):
        graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


res = {"state": f"{B} x {D} fp32", "unit": "us per trial step (6 combines + 7 GEMMs + error norm), hipGraph replay"}
ref = None
ARRANGEMENTS = (("serial", 1, False), ("split2", 2, False), ("split2_skew", 2, True),
                ("split4", 4, False), ("split4_skew", 4, True), ("split8_skew", 8, True),
                ("seq2", 2, "seq"), ("seq4", 4, "seq"), ("seq8", 8, "seq"))
if len(sys.argv) > 1:          # one arrangement only (for a rocprofv3 kernel trace of it: do the chains really overlap?)
    ARRANGEMENTS = tuple(a for a in ARRANGEMENTS if a[0] in sys.argv[1:])
for name, n_parts, skew in ARRANGEMENTS:
    try:
        with torch.no_grad():
            graph, keep, parts = capture(n_parts, skew)
            us = time_graph(graph)
        y1 = torch.cat([k[0] for k in keep])
        if ref is None:
            ref = y1.clone()
        res[name] = {"us_per_step": us, "stages_per_s": 6e6 / us, "same_values_as_serial": bool(torch.equal(y1, ref))}
        del graph, keep, parts
    except Exception as exc:
        res[name] = {"error": repr(exc)}
    torch.cuda.synchronize()
print(json.dumps(res, indent=1))
