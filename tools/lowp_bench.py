#!/usr/bin/env python
"""bf16 / fp16 states (r05): bench.py's `low_precision` extras object on its own —

    python tools/lowp_bench.py > gpurun_out/r05_lowp_bench.json

dopri5 trial steps of the 65536 x 128 workload with a reduced-precision STATE on the HIP kernels of
csrc/tdeq_kernels_lp.hpp vs the package's torch-op host path (what r04 ran for such states) on the same MI355X, and the HBM
rate of the 16-bit stage combine."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    print(json.dumps(bench.low_precision_regime(torch.device("cuda", 0)), indent=1))
