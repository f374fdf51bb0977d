"""Run by tests/test_dist_gpu.py in a subprocess on the GPU box: the sharded-solve collectives through RCCL
(backend "nccl") on device buffers, at world size 1 with the process group given explicitly (torchdiffeq_amd.dist
honours an explicit group at world size 1, so nothing is short-circuited):

  * odeint_adjoint_sharded            -> adjoint._allreduce_tail: ONE all-reduce on the contiguous parameter-adjoint
                                         tail of the flat augmented state (+ the time gradients)
  * odeint_adjoint_sharded(sync_steps) -> solvers._LockStep (norm sums, on device) + per-evaluation VJP all-reduce
  * odeint_sharded(sync_steps)         -> solvers._LockStep in the forward solve

A sum over one rank is the identity, so every result must equal the plain single-process call bit for bit."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchdiffeq_amd as tda  # noqa: E402
from torchdiffeq_amd import dist as tdist  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29631")
    torch.cuda.set_device(0)
    # the same call torchdiffeq_amd.dist.init_from_env makes for backend nccl: communicator bound to this rank's GPU
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    dist.barrier()
    dev = torch.device("cuda:0")
    calls = {"n": 0, "cuda": 0, "bytes": 0}
    orig = dist.all_reduce

    def counting(tensor, *a, **kw):
        calls["n"] += 1
        calls["cuda"] += int(tensor.is_cuda)
        calls["bytes"] += tensor.numel() * tensor.element_size()
        return orig(tensor, *a, **kw)
    dist.all_reduce = counting

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 16)).to(dev)

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = net

        def forward(self, t, y):
            return self.net(y) * torch.cos(t)

    f = F()
    y0 = torch.randn(512, 16, device=dev)
    t = torch.tensor([0.0, 0.4, 1.0], device=dev)

    def grads(fn, **kw):
        for p in f.parameters():
            p.grad = None
        x = y0.clone().requires_grad_(True)
        tt = t.clone().requires_grad_(True)
        y = fn(f, x, tt, rtol=1e-5, atol=1e-7, method="dopri5", **kw)
        (y[-1].pow(2).sum() + y[1].sum()).backward()
        return [y.detach(), x.grad, tt.grad] + [p.grad.clone() for p in f.parameters()]

    base = grads(tda.odeint_adjoint)
    assert calls["n"] == 0
    # (1) ONE all-reduce for everything `backward` sums over the batch — the parameter-adjoint tail and, in the same
    # buffer, the len(t) time gradients (reference adjoint.py:121-153) — on a device buffer, via RCCL
    sharded = grads(tdist.odeint_adjoint_sharded, group=dist.group.WORLD)
    assert calls["n"] == 1 and calls["cuda"] == 1, calls
    n_params = sum(p.numel() for p in f.parameters())
    assert calls["bytes"] >= 4 * n_params, calls
    for a, b in zip(base, sharded):
        assert torch.equal(a, b), "odeint_adjoint_sharded differs from odeint_adjoint at world size 1"
    # (2) lock step: norm sums all-reduced per trial step on the device, VJPs per evaluation
    n0 = calls["n"]
    lock = grads(tdist.odeint_adjoint_sharded, group=dist.group.WORLD, sync_steps=True)
    assert calls["n"] - n0 > 10 and calls["cuda"] == calls["n"], calls
    for a, b in zip(base, lock):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), "lock-step adjoint differs at world size 1"
    # (2b) the same with the host-side reduction (TDEQ_LOOKAHEAD=0: sums -> host -> tensor -> all_reduce): the device-
    #      resident path above (finalize -> RCCL all-reduce on the device buffer -> tdeq_step_controller) must take the
    #      same steps
    os.environ["TDEQ_LOOKAHEAD"] = "0"
    lock_host = grads(tdist.odeint_adjoint_sharded, group=dist.group.WORLD, sync_steps=True)
    os.environ.pop("TDEQ_LOOKAHEAD")
    for a, b in zip(lock, lock_host):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), "device-resident and host-side lock step differ"
    # (3) forward-only lock step
    with torch.no_grad():
        y_plain = tda.odeint(f, y0, t, rtol=1e-6, atol=1e-8)
        n0 = calls["n"]
        y_lock = tdist.odeint_sharded(f, y0, t, group=dist.group.WORLD, sync_steps=True, rtol=1e-6, atol=1e-8)
    assert calls["n"] > n0
    assert torch.allclose(y_plain, y_lock, rtol=1e-5, atol=1e-6)
    # (4) captured trial steps in a process that HAS a live RCCL communicator: its watchdog thread issues HIP calls of
    #     its own, which a "global" capture would be invalidated by (_graph._capture uses thread_local); a failed
    #     capture would fall back to the eager path with a warning — turned into an error here
    import warnings
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        y_eager = tda.odeint(f, y0, t, rtol=1e-6, atol=1e-8)
        for _ in range(3):
            y_graph = tda.odeint(f, y0, t, rtol=1e-6, atol=1e-8, options=dict(hip_graph=True))
            assert torch.equal(y_eager, y_graph)
    x = torch.ones(8, device=dev)
    dist.all_reduce(x)              # ... and the communicator still works after the captures
    assert float(x.sum()) == 8.0
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL_WORLD1_OK", calls)


if __name__ == "__main__":
    main()
