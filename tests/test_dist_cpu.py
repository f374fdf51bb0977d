"""world_size-2 gloo tests of the sharded solves.
  * default mode: parameter gradients summed by ONE all-reduce equal the single-process gradients over the whole
    batch within the solve tolerance; y0 gradients stay sharded.
  * lock-step mode (sync_steps=True): the shards share the whole-batch step controller, so the forward rows, the
    number of func evaluations and the gradients equal the single-process run to rounding."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(dtype):
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 6)).to(dtype)

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = net

        def forward(self, t, y):
            return self.net(y) * torch.cos(t)

    g = torch.Generator().manual_seed(1)
    y0 = torch.randn(11, 6, generator=g, dtype=torch.float64).to(dtype)     # 11 rows: uneven shards
    return F(), y0


def _patch_backend():
    from oracle.kernels import OracleKernels
    from torchdiffeq_amd import _native
    ok = OracleKernels()
    _native.get_kernels = lambda device, dtype=None: ok      # host-logic test on CPU tensors (see conftest.cpu_backend)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    if os.environ.get("TDEQ_TEST_HOST_PATH") == "1":
        import warnings
        warnings.simplefilter("ignore")      # the package's own torch-op host path for CPU states (HostPathWarning)
    else:
        _patch_backend()
    from torchdiffeq_amd import dist as tdist
    r, w, _ = tdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    f, y0 = _make(torch.float64)
    t = torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64, requires_grad=True)
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(a[0].numel()), orig(*a, **k))[1]
    shard = tdist.shard_batch(y0).clone().requires_grad_(True)
    y = tdist.odeint_adjoint_sharded(f, shard, t, rtol=1e-9, atol=1e-11)
    (y[-1].pow(2).sum() + y[1].sum()).backward()
    dist.all_reduce = orig
    torch.save(dict(gy=shard.grad, gp=[p.grad for p in f.parameters()], gt=t.grad, calls=calls,
                    rows=tdist.shard_rows(11, rank, world)), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_adjoint_world2(tmp_path, cpu_backend):
    import torchdiffeq_amd as tda
    world = 2
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=False) for r in range(world)]

    f, y0 = _make(torch.float64)
    y0 = y0.clone().requires_grad_(True)
    t = torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64, requires_grad=True)
    y = tda.odeint_adjoint(f, y0, t, rtol=1e-9, atol=1e-11)
    (y[-1].pow(2).sum() + y[1].sum()).backward()
    # per-shard step controllers take slightly different steps than the whole-batch solve
    # (SURVEY.md §8e "caveat for parity"): compare within the solve tolerance, not bitwise.
    for r in range(world):
        assert torch.allclose(res[r]["gy"], y0.grad[res[r]["rows"]], rtol=1e-6, atol=1e-8)
        for g_shard, p in zip(res[r]["gp"], f.parameters()):
            assert torch.allclose(g_shard, p.grad, rtol=1e-6, atol=1e-8)
        assert torch.allclose(res[r]["gt"], t.grad, rtol=1e-6, atol=1e-8)
    # identical (already reduced) parameter gradients on both ranks
    for a, b in zip(res[0]["gp"], res[1]["gp"]):
        assert torch.equal(a, b)
    # exactly ONE all-reduce per backward: the parameter tail and, in the same buffer, the len(t) = 3 entries of dL/dt
    # (t.requires_grad) — SURVEY.md §8e, reference adjoint.py:121-153
    n_tail = sum(-(-p.numel() // 1024) * 1024 for p in f.parameters())
    assert len(res[0]["calls"]) == 1 and res[0]["calls"][0] == n_tail + 3, res[0]["calls"]
    assert res[0]["rows"] == slice(0, 6) and res[1]["rows"] == slice(6, 11)


def test_sharded_adjoint_world2_on_the_host_path(tmp_path, monkeypatch):
    """The same two-rank sharded adjoint with NO test backend substituted: CPU shards are integrated by the package's
    torch-op host path (r03, _fallback.HostKernels) and the parameter adjoints are summed by the one gloo all-reduce.
    Must equal the single-process solve on the same path."""
    import warnings
    import torchdiffeq_amd as tda
    monkeypatch.setenv("TDEQ_TEST_HOST_PATH", "1")
    world = 2
    port = 29900 + (os.getpid() % 90)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=False) for r in range(world)]
    f, y0 = _make(torch.float64)
    y0 = y0.clone().requires_grad_(True)
    t = torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64, requires_grad=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = tda.odeint_adjoint(f, y0, t, rtol=1e-9, atol=1e-11)
        (y[-1].pow(2).sum() + y[1].sum()).backward()
    for r in range(world):
        assert torch.allclose(res[r]["gy"], y0.grad[res[r]["rows"]], rtol=1e-6, atol=1e-8)
        for g_shard, p in zip(res[r]["gp"], f.parameters()):
            assert torch.allclose(g_shard, p.grad, rtol=1e-6, atol=1e-8)
        assert torch.allclose(res[r]["gt"], t.grad, rtol=1e-6, atol=1e-8)
    assert len(res[0]["calls"]) == 1


class _CountingModule(torch.nn.Module):
    def __init__(self, inner):
        super().__init__()
        self.inner, self.nfe = inner, 0

    def forward(self, t, y):
        self.nfe += 1
        return self.inner(t, y)


def _lockstep_run(tda_odeint_adjoint, f, y0, t, **kw):
    y = tda_odeint_adjoint(f, y0, t, rtol=1e-8, atol=1e-10, method="dopri5", **kw)
    nfe_fwd = f.nfe
    (y[-1].pow(2).sum() + y[1].sum()).backward()
    return y.detach(), nfe_fwd, f.nfe - nfe_fwd


def _lockstep_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    _patch_backend()
    from torchdiffeq_amd import dist as tdist
    tdist.init_from_env(backend="gloo")
    f, y0 = _make(torch.float64)
    f = _CountingModule(f)
    t = torch.tensor([0.0, 0.4, 1.0], dtype=torch.float64, requires_grad=True)
    shard = tdist.shard_batch(y0).clone().requires_grad_(True)
    # forward only, lock step
    with torch.no_grad():
        y_fwd = tdist.odeint_sharded(f, shard.detach(), t.detach(), rtol=1e-8, atol=1e-10, method="dopri5")
    nfe_plain = f.nfe
    f.nfe = 0
    y, nfe_fwd, nfe_bwd = _lockstep_run(lambda *a, **k: tdist.odeint_adjoint_sharded(*a, sync_steps=True, **k),
                                        f, shard, t)
    torch.save(dict(y_fwd=y_fwd, y=y, nfe_plain=nfe_plain, nfe_fwd=nfe_fwd, nfe_bwd=nfe_bwd, gy=shard.grad,
                    gp=[p.grad for p in f.parameters()], gt=t.grad, rows=tdist.shard_rows(11, rank, world)),
               os.path.join(out_dir, f"ls{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_lockstep_world2_equals_single_process(tmp_path, cpu_backend):
    import torchdiffeq_amd as tda
    world = 2
    port = 29950 + (os.getpid() % 300)
    mp.spawn(_lockstep_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"ls{r}.pt"), weights_only=False) for r in range(world)]

    f, y0 = _make(torch.float64)
    f = _CountingModule(f)
    y0 = y0.clone().requires_grad_(True)
    t = torch.tensor([0.0, 0.4, 1.0], dtype=torch.float64, requires_grad=True)
    y, nfe_fwd, nfe_bwd = _lockstep_run(tda.odeint_adjoint, f, y0, t)
    for r in range(world):
        rows = res[r]["rows"]
        # same step sequence as the whole-batch solve: same NFE, rows equal to rounding (the cross-rank sum of the
        # error norm is associated differently, which moves dt by ~1e-16 relative)
        assert res[r]["nfe_plain"] == nfe_fwd and res[r]["nfe_fwd"] == nfe_fwd and res[r]["nfe_bwd"] == nfe_bwd
        assert torch.allclose(res[r]["y_fwd"], y[:, rows], rtol=1e-13, atol=1e-14)
        assert torch.allclose(res[r]["y"], y[:, rows], rtol=1e-13, atol=1e-14)
        assert torch.allclose(res[r]["gy"], y0.grad[rows], rtol=1e-11, atol=1e-13)
        for g_shard, p in zip(res[r]["gp"], f.parameters()):
            assert torch.allclose(g_shard, p.grad, rtol=1e-11, atol=1e-13)
        assert torch.allclose(res[r]["gt"], t.grad, rtol=1e-11, atol=1e-13)
    for a, b in zip(res[0]["gp"], res[1]["gp"]):
        assert torch.equal(a, b)


def _adams_lockstep_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    _patch_backend()
    from torchdiffeq_amd import dist as tdist
    tdist.init_from_env(backend="gloo")
    f, y0 = _make(torch.float64)
    f = _CountingModule(f)
    t = torch.linspace(0, 1, 21, dtype=torch.float64)
    shard = tdist.shard_batch(y0).clone()
    with torch.no_grad():
        y = tdist.odeint_sharded(f, shard, t, rtol=1e-7, atol=1e-9, method="implicit_adams")
    torch.save(dict(y=y, nfe=f.nfe, rows=tdist.shard_rows(11, rank, world)), os.path.join(out_dir, f"ad{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_implicit_adams_lockstep_world2_equals_single_process(tmp_path, cpu_backend):
    """The corrector's convergence census is all-reduced in lock-step mode: every shard runs the whole-batch number
    of iterations per step, so rows and evaluation counts equal the single-process solve EXACTLY (the census is an
    integer count — no rounding in the cross-rank sum)."""
    import torchdiffeq_amd as tda
    world = 2
    port = 30300 + (os.getpid() % 300)
    mp.spawn(_adams_lockstep_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"ad{r}.pt"), weights_only=False) for r in range(world)]
    f, y0 = _make(torch.float64)
    f = _CountingModule(f)
    with torch.no_grad():
        y = tda.odeint(f, y0, torch.linspace(0, 1, 21, dtype=torch.float64), rtol=1e-7, atol=1e-9,
                       method="implicit_adams")
    for r in range(world):
        assert res[r]["nfe"] == f.nfe
        assert torch.equal(res[r]["y"], y[:, res[r]["rows"]])


def test_shard_rows_cover_batch():
    from torchdiffeq_amd.dist import shard_rows
    for n in (1, 7, 8, 65536):
        for world in (1, 2, 3, 8):
            rows = [shard_rows(n, r, world) for r in range(world)]
            assert rows[0].start == 0 and rows[-1].stop == n
            assert all(a.stop == b.start for a, b in zip(rows, rows[1:]))
            sizes = [s.stop - s.start for s in rows]
            assert max(sizes) - min(sizes) <= 1
