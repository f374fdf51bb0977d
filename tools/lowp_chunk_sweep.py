#!/usr/bin/env python
"""bf16 dopri5 trial step vs the norm kernels' chunk size (TDEQ_CHUNK, elements per workgroup of the norm launches): run as
    for c in 2048 4096 8192; do TDEQ_CHUNK=$c python tools/lowp_chunk_sweep.py; done"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    out = {"TDEQ_CHUNK": os.environ.get("TDEQ_CHUNK", "default")}
    for name, graph in (("eager", False), ("captured", True)):
        r = bench.lowp_steps(torch.bfloat16, "hip", 40, 10, dev, hip_graph=graph)
        out[name] = r["ms_per_step"]
        if r["breakdown"]:
            out["kernels"] = {k[:70]: round(v["avg_us"], 2) for k, v in r["breakdown"]["top_kernels"].items()}
    print(json.dumps(out))
