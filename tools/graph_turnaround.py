"""Where do the microseconds of a captured trial step go (8192 x 128 shard of cfg2)?  (a) the solver's own loop (replay ->
poll the controller's words -> host bookkeeping -> next replay), (b) the same graphs replayed back to back with no
read-back in between (what the GPU alone needs), (c) replay + one poll and nothing else.  (a) - (b) = the host
turn-around per step."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
A, y0 = bench.make_problem(dev, rows=slice(0, rows))
At = A.T.contiguous()
solver = bench.make_stepper(lambda t, y: y @ At, y0, hip_graph=True)
res = {"state": f"{rows} x 128 fp32"}
with torch.no_grad():
    for _ in range(30):
        solver._trial_step()
    N = 300

    def timed(fn):
        out = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(N):
                fn()
            torch.cuda.synchronize()
            out.append(1e6 * (time.perf_counter() - t0) / N)
        return sorted(out)[2]
    res["solver_loop_us"] = timed(solver._trial_step)
    g = solver._g
    kern = solver.kernels
    sides = [0]

    def back_to_back():
        g.graphs[sides[0]].replay()
        sides[0] ^= 1
    res["back_to_back_replays_us"] = timed(back_to_back)

    def replay_and_poll():
        kern.arm_readback(solver.plan)
        g.graphs[sides[0]].replay()
        sides[0] ^= 1
        kern.read_ctrl(solver.plan)
    res["replay_then_poll_us"] = timed(replay_and_poll)

    def replay_and_sync():
        g.graphs[sides[0]].replay()
        sides[0] ^= 1
        torch.cuda.current_stream().synchronize()
    res["replay_then_stream_sync_us"] = timed(replay_and_sync)
    t0 = time.perf_counter()
    for _ in range(2000):
        kern.arm_readback(solver.plan)
    res["arm_readback_host_us"] = 1e6 * (time.perf_counter() - t0) / 2000
solver.plan.expect = ()
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "graph_turnaround.json"), "w"), indent=1)
