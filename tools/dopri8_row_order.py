"""r06: does summing the 14-slot dopri8 rows in the association ATen's CPU `torch.sum` uses remove the
first-step residue against the reference?  Build container only (imports /root/reference).

The reference forms every tableau row as `torch.sum(k[..., :L] * (coef * dt), dim=-1)` (torchdiffeq/_impl/rk_common.py:79,
89): the products — zeros included — are rounded in T, then added by ATen's inner-contiguous CPU kernel.  Its association,
determined empirically and reproduced bit for bit for every L = 1..14 in fp32 and fp64 (`aten_sum` below; SumKernel.cpp's
`vectorized_inner_sum` for L >= V = 8 floats / 4 doubles: V lane sums over the full vectors, then `0 + tail elements in
order + lanes in order`; `scalar_inner_sum` below V: four interleaved partial sums, the tail added to the first, then
p0 + p1 + p2 + p3).  This package's kernels add the NON-ZERO products left to right.

Three variants of the CPU oracle's step arithmetic (oracle/reference_solver.py, same host logic, func evaluated by torch on
the CPU so that it is bit-identical on both sides) against the imported reference on small dopri8 solves:
    left_to_right   the shipped order
    aten_error_row  ATen's order for the error row only (what a `TDEQ_ERRROW=aten` kernel variant would do)
    aten_all_rows   ATen's order for every stage row, the solution row and the error row
Reported per variant and dtype: cases whose accepted step sequence equals the reference's to 1e-12 / worst relative step
difference / worst solution difference.

    PYTHONDONTWRITEBYTECODE=1 python tools/dopri8_row_order.py [seed] [cases]"""
import json
import os
import random
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import torchdiffeq as ref  # noqa: E402
from oracle import reference_solver as orc  # noqa: E402

torch.set_num_threads(1)


def aten_sum(x):
    """x: [n, L] products (zeros in place) -> torch.sum(x, -1) as ATen's CPU kernel associates it."""
    n, L = x.shape
    dt = x.dtype
    V = 8 if dt == np.float32 else 4
    add = lambda a, b: (a + b).astype(dt)
    if L >= V:
        nv = L // V
        lanes = [x[:, l].copy() for l in range(V)]
        for c in range(1, nv):
            for l in range(V):
                lanes[l] = add(lanes[l], x[:, c * V + l])
        acc = None
        for k in range(nv * V, L):
            acc = x[:, k].copy() if acc is None else add(acc, x[:, k])
        for l in range(V):
            acc = lanes[l] if acc is None else add(acc, lanes[l])
        return acc
    p = [None] * 4
    for i in range(L // 4):
        for k in range(4):
            p[k] = x[:, 4 * i + k].copy() if p[k] is None else add(p[k], x[:, 4 * i + k])
    for i in range(4 * (L // 4), L):
        p[0] = x[:, i].copy() if p[0] is None else add(p[0], x[:, i])
    acc = p[0]
    for k in range(1, 4):
        if p[k] is not None:
            acc = add(acc, p[k])
    return acc


def check_model():
    for dt in (torch.float32, torch.float64):
        for L in range(1, 15):
            g = torch.Generator().manual_seed(L)
            k = torch.randn(2003, 14, generator=g, dtype=torch.float64).to(dt)
            c = (torch.randn(L, generator=g, dtype=torch.float64) * torch.logspace(-3, 3, L, dtype=torch.float64)).to(dt)
            prod = k[..., :L] * c
            assert (aten_sum(prod.numpy()) == torch.sum(prod, dim=-1).numpy()).all(), (dt, L)


class AtenOps(orc.NumpyOps):
    """Rows summed like ATen: `rows` = which of them ('error' or 'all')."""

    def __init__(self, rows):
        self.rows = rows

    def _sum(self, ks, coef, dt, dtype):
        c = [dtype.type(cj) * dtype.type(dt) for cj in coef]          # (beta_i * dt): a T vector
        return aten_sum(np.stack([k * cj for k, cj in zip(ks, c)], axis=1))

    def combine(self, y0, ks, coef, dt):
        if self.rows != "all" or len(ks) == 1:
            return super().combine(y0, ks, coef, dt)
        return y0 + self._sum(ks, coef, dt, y0.dtype)

    def error_ratio_sq(self, y0, y1, ks, coef, dt, rtol, atol, segments):
        T = y0.dtype.type
        err = self._sum(ks, coef, dt, y0.dtype)
        out = []
        for (lo, hi), r_, a_ in zip(segments, rtol, atol):
            tol = T(a_) + T(r_) * np.fmax(np.abs(y0[lo:hi]), np.abs(y1[lo:hi]))
            r = (err[lo:hi] / tol).astype(np.float64)
            out.append(float(np.dot(r, r)) / max(hi - lo, 1))
        return out, not (np.isfinite(y0).all() and np.isfinite(y1).all())


def dense_nz(row):
    return list(range(len(row))), [float(v) for v in row]


def sparse_nz(row):
    idx = [j for j, v in enumerate(row) if v != 0.0]
    return idx, [float(row[j]) for j in idx]


class Mixed:
    """`_nz` replacement: dense rows (zeros in place) where the ATen order is wanted, the non-zeros elsewhere."""

    def __init__(self, rows, tab):
        self.rows, self.err = rows, tab.c_error

    def __call__(self, row):
        if self.rows == "all" or (self.rows == "error" and row is self.err):
            return dense_nz(row)
        return sparse_nz(row)


def run_case(rng, dtype):
    g = torch.Generator().manual_seed(rng.randrange(10 ** 6))
    d = rng.choice([2, 3, 5])
    n = rng.choice([1, 4, 9])
    W = (torch.randn(d, d, generator=g, dtype=torch.float64) * 0.6).to(dtype)
    y0 = torch.randn(n, d, generator=g, dtype=torch.float64).to(dtype)
    kind = rng.choice(["tanh", "linear_t", "cubic"])
    t1 = rng.uniform(0.5, 2.0)
    t = torch.tensor([0.0, t1], dtype=dtype)
    rtol, atol = (1e-5, 1e-7) if dtype == torch.float32 else (1e-9, 1e-11)

    def f(tt, y):
        if kind == "tanh":
            return torch.tanh(y @ W.T) * torch.cos(tt)
        if kind == "linear_t":
            return y @ W.T * (1 + 0.3 * tt) - 0.2 * y
        return -0.3 * y ** 3 + torch.sin(3 * tt) * (y @ W.T)
    acc = []

    class F(torch.nn.Module):
        def forward(self, tt, y):
            return f(tt, y)

        def callback_accept_step(self, t0, y_, dt):
            acc.append(float(dt))
    with torch.no_grad():
        y_ref = ref.odeint(F(), y0, t, method="dopri8", rtol=rtol, atol=atol)[-1].numpy()
    out = {}
    npdt = np.float32 if dtype == torch.float32 else np.float64

    def fn(tt, yflat):
        with torch.no_grad():
            return f(torch.tensor(tt, dtype=dtype), torch.from_numpy(yflat.reshape(n, d))).numpy().reshape(-1)
    for name, rows in (("left_to_right", None), ("aten_error_row", "error"), ("aten_all_rows", "all")):
        tab = orc.tableau("dopri8")
        solver = orc.AdaptiveRK(fn, y0.numpy().reshape(-1).astype(npdt), tab, rtol, atol,
                                ops=orc.NumpyOps() if rows is None else AtenOps(rows))
        saved = orc._nz
        orc._nz = sparse_nz if rows is None else Mixed(rows, tab)
        try:
            solver.before_integrate(0.0)
            y = solver.advance(float(t[1]))
        finally:
            orc._nz = saved
        dts = solver.dts
        same_len = len(dts) == len(acc)
        m = min(len(dts), len(acc))
        ddt = max(abs(a - b) / abs(b) for a, b in zip(dts[:m], acc[:m])) if m else 0.0
        out[name] = (same_len, ddt, float(np.abs(y.reshape(n, d) - y_ref).max() / np.abs(y_ref).max()))
    return out


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    check_model()
    rng = random.Random(seed)
    report = {"seed": seed, "cases_per_dtype": cases, "aten_sum_model": "bit-exact for L = 1..14, fp32 and fp64",
              "cpu_capability": torch.backends.cpu.get_cpu_capability()}
    for dtype in (torch.float32, torch.float64):
        agg = {}
        for _ in range(cases):
            for name, (same_len, ddt, dy) in run_case(rng, dtype).items():
                a = agg.setdefault(name, {"identical_step_sequences": 0, "different_step_count": 0, "worst_dt_rel": 0.0,
                                          "worst_solution_rel": 0.0})
                a["identical_step_sequences"] += int(same_len and ddt <= 1e-12)
                a["different_step_count"] += int(not same_len)
                a["worst_dt_rel"] = max(a["worst_dt_rel"], ddt)
                a["worst_solution_rel"] = max(a["worst_solution_rel"], dy)
        report[str(dtype)[6:]] = agg
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
