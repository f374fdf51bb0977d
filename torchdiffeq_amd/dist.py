"""Batch-axis data parallelism for the RK hot path: one process per GPU, RCCL over xGMI.

The solver arithmetic is elementwise over the state, so independent batch rows shard across ranks with
NO data-path collective: each rank integrates its rows with its own accept/reject loop.
The only cross-rank quantity is the adjoint's parameter gradient (a sum over the batch): the θ-adjoint
segments are a contiguous tail of the flat augmented state, summed with ONE all-reduce at the end of
`backward` (≈0.4 MB for cfg3 — latency-bound on xGMI, so a single coalesced call, not a bucketed ring).

Optional lock-step mode (`sync_steps=True`): the shards share the whole-batch step controller through one tiny
all-reduce of the error sums per trial step (and, in the adjoint, of the batch-summed VJPs per evaluation), which
reproduces the single-device step sequence and results (SURVEY.md §8e "exact mode").

Nothing of this exists in the reference (no torch.distributed call anywhere in its tree).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from .adjoint import odeint_adjoint
from .odeint import odeint


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (as set by
    `python -m torch.distributed.run`).  backend: 'nccl' (= RCCL on ROCm) when a GPU is visible, else
    'gloo'.  Returns (rank, world_size, local_rank); a no-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (TDEQ_DIST_FORCE_INIT=1: create the process group at world size 1 too — how bench.py's communicator census is
    # exercised through RCCL on a one-GPU box)
    if (world > 1 or os.environ.get("TDEQ_DIST_FORCE_INIT") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            # rank r owns GPU r: make it the current device AND bind the communicator to it (no guessing in barrier(),
            # the RCCL communicator is created eagerly — a wrong mapping fails here, not in the first collective)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_rows(n_rows: int, rank: int, world: int) -> slice:
    """Contiguous row block of `rank` (block sizes differ by at most one row)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return slice(lo, lo + base + (1 if rank < rem else 0))


def shard_batch(y0: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    """This rank's rows of a batch-first state."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    return y0[shard_rows(y0.shape[0], rank, world)]


def odeint_sharded(func, y0_shard, t, *, group=None, sync_steps=True, **kwargs):
    """`odeint` on this rank's batch shard.  sync_steps=False: every rank runs its own accept/reject loop (no
    communication at all; results agree with the whole-batch solve within the tolerances).  sync_steps=True
    ("lock step"): the per-segment error sums are added over the ranks — one all-reduce of 3·n_seg doubles per
    trial step, latency-bound on xGMI — so every shard takes exactly the step sequence of the whole-batch solve
    and its rows come out as in a single-device run (adaptive methods; fixed grids are in lock step anyway)."""
    if sync_steps and dist.is_initialized() and (dist.get_world_size(group) > 1 or group is not None):
        options = dict(kwargs.pop("options", None) or {})
        options["dist_sync"] = True if group is None else group
        kwargs["options"] = options
    return odeint(func, y0_shard, t, **kwargs)


def odeint_adjoint_sharded(func, y0_shard, t, *, group=None, sync_steps=False, **kwargs):
    """`odeint_adjoint` on this rank's batch shard; parameter gradients (and dL/dt, if requested) come out
    of `backward` already summed over the ranks of `group` (default group if None).  Gradients wrt y0 stay
    sharded.  With no initialised process group this is plain `odeint_adjoint`.

    sync_steps=False (default): per-shard step controllers, ONE all-reduce of the parameter adjoints at the end of
    `backward`.  sync_steps=True: lock step — forward as `odeint_sharded`; in the backward solve the time- and
    parameter-VJPs are all-reduced at every evaluation (1 + P words), which makes every rank integrate the
    whole-batch adjoint system with the whole-batch step sequence: gradients equal the single-device ones to
    rounding, at the price of one small collective per evaluation and per trial step."""
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or group is not None):
        # (an explicitly given group is honoured at world size 1 too: the collectives then run through the backend —
        # how the RCCL path is exercised on a one-GPU box)
        g = True if group is None else group
        extra = {"dist_sync": g} if sync_steps else {"dist_group": g}
        user_options = kwargs.get("options")
        if sync_steps:
            options = dict(user_options or {})
            options["dist_sync"] = g
            kwargs["options"] = options
        if kwargs.get("adjoint_options") is not None:
            kwargs["adjoint_options"] = {**kwargs["adjoint_options"], **extra}
        elif user_options is None:
            kwargs["adjoint_options"] = dict(extra)       # nothing for odeint_adjoint to infer from
        else:
            # odeint_adjoint infers (and validates) adjoint_options from `options` as always; the dist keys are
            # merged in after that, so its ValueError for `adjoint_method != method` with `options` still fires
            kwargs["_adjoint_extra"] = extra
    return odeint_adjoint(func, y0_shard, t, **kwargs)
