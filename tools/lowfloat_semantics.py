"""How ATen's CPU kernels round a Python number / 0-dim tensor next to a bf16 / fp16 / fp32 tensor — the measurements
`torchdiffeq_amd/_scalars.py` is written from.  Compares the host scalar classes with torch on random operands and prints
the mismatch counts (all zero = the emulation is exact for these operations).  Usage: python tools/lowfloat_semantics.py"""
import collections
import math
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchdiffeq_amd._scalars import BFloat16Scalar, Float16Scalar, nextafter, power, rdiv  # noqa: E402


def same(x, y):
    x, y = float(x), float(y)
    return x == y or (math.isnan(x) and math.isnan(y))


def main(n=4000):
    random.seed(0)
    bad = collections.Counter()
    ops = (("add", lambda x, y: x + y), ("sub", lambda x, y: x - y), ("mul", lambda x, y: x * y),
           ("div", lambda x, y: x / y))
    for cls, dt in ((BFloat16Scalar, torch.bfloat16), (Float16Scalar, torch.float16)):
        for _ in range(n):
            a = random.uniform(-1, 1) * 10 ** random.uniform(-3, 2)
            b = random.uniform(-1, 1) * 10 ** random.uniform(-3, 2)
            A, B = cls(a), cls(b)
            ta, tb = torch.tensor(a, dtype=dt), torch.tensor(b, dtype=dt)
            bad[(dt, "cast")] += not same(A, ta)
            for name, f in ops:
                bad[(dt, name, "low,low")] += not same(f(A, B), f(ta, tb))
                bad[(dt, name, "low,py")] += not same(f(A, b), f(ta, b))
                bad[(dt, name, "py,low")] += not same(f(a, B), f(a, tb))
                bad[(dt, name, "low,f64")] += not same(f(A, np.float64(b)), f(ta, torch.tensor(b, dtype=torch.float64)))
            for e in (0.2, 1 / 3, 0.125, 0.5):
                bad[(dt, "pow", e)] += not same(abs(A) ** e, ta.abs() ** e)
            if dt is torch.bfloat16:        # torch.nextafter has no Half kernel on the CPU
                for tgt in (A + 1, A - 1, B):
                    bad[(dt, "nextafter")] += not same(nextafter(A, tgt), torch.nextafter(ta, torch.tensor(float(tgt), dtype=dt)))
    for T, dt in ((np.float32, torch.float32), (np.float64, torch.float64)):
        for _ in range(n):
            a = abs(random.uniform(-1, 1) * 10 ** random.uniform(-3, 2))
            ta, fa = torch.tensor(a, dtype=dt), T(a)
            bad[(dt, "0.01/x")] += not same(rdiv(0.01, fa), 0.01 / ta)
            for e in (0.2, 1 / 3, 0.125, 0.5):
                bad[(dt, "pow", e)] += not same(power(fa, e), ta ** e)
    for k in sorted(bad, key=str):
        print(k, bad[k])
    return sum(bad.values())


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
