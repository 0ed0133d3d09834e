"""
Element-axis sharding of the DG einsums across the GPUs of one node.

New functionality (the reference is single process / single queue:
``src/feinsum/measure.py:113,201``).  Every output entry depends only on inputs
of the same element plus the small replicated operator, so the path shards
trivially: one process per GPU (``torch.distributed``, backend
``nccl`` = RCCL over xGMI on ROCm, ``gloo`` on CPU for tests), a contiguous
block of elements per rank, NO collective on the data path.  The only exchange
is an all-gather of each shard's *result reduction* (a few float64 per output:
sum, sum of squares, max |.|), which is what north_star asks for; gathering
whole fields would cost ~20x the compute time over xGMI and is
available separately as :func:`allgather_field`.
"""

from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, List, Mapping, Sequence, Tuple

TILE = 16   # elements per MFMA wave tile: shard boundaries are tile aligned


def shard_bounds(E: int, world_size: int, rank: int, align: int = TILE) -> Tuple[int, int]:
    """
    Contiguous block split of ``range(E)``: every rank gets ``floor(E / world /
    align) * align`` elements, the remainder goes to the last rank, so all shard
    starts are multiples of *align* (full MFMA tiles; only the last rank can
    have a ragged tail).
    """
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} for world size {world_size}")
    if E < 0:
        raise ValueError("E must be >= 0")
    per = (E // world_size) // align * align
    start = rank * per
    stop = E if rank == world_size - 1 else start + per
    return start, stop


def slice_long_axis(arr: Any, axis: int, start: int, stop: int) -> Any:
    """``arr`` restricted to ``[start, stop)`` along *axis*, as a dense copy
    (each rank keeps its shard dense: J is (3,3,E), grad out (3,E,Np), ...)."""
    idx = [slice(None)] * arr.ndim
    idx[axis] = slice(start, stop)
    sl = arr[tuple(idx)]
    return sl.contiguous() if hasattr(sl, "contiguous") else sl.copy()


def shard_host_arrays(einsum: Any, host: Mapping[str, Any], world_size: int, rank: int) -> dict:
    """Shard every array that carries a ``SizeParam`` axis; replicate the rest."""
    from feinsum_amd.einsum import SizeParam

    out = {}
    for name, shape in einsum.arg_to_shape.items():
        arr = host[name]
        for axis, d in enumerate(shape):
            if isinstance(d, SizeParam):
                s, e = shard_bounds(arr.shape[axis], world_size, rank)
                arr = slice_long_axis(arr, axis, s, e)
        out[name] = arr
    return out


@dataclass(frozen=True)
class DistInfo:
    rank: int
    local_rank: int
    world_size: int
    backend: str


def init_distributed(backend: str | None = None) -> DistInfo:
    """
    Join the process group described by RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT (torch.distributed.run sets them); a plain
    ``python bench.py`` is world size 1 and joins nothing.
    """
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if backend is None:
        # FEINSUM_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsals; RCCL refuses that)
        backend = os.environ.get("FEINSUM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        local_rank %= max(torch.cuda.device_count(), 1)
    # FEINSUM_DIST_FORCE=1 joins a group even at world size 1: every collective of the sharded path then runs through
    # the backend (a one-GPU box can rehearse the RCCL calls themselves that way)
    if world > 1 or os.environ.get("FEINSUM_DIST_FORCE") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        if not dist.is_initialized():
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return DistInfo(rank, local_rank, world, backend)


def in_group() -> bool:
    """Has this process joined a process group (then the exchanges below are real collectives)?"""
    import torch.distributed as dist

    return bool(dist.is_available() and dist.is_initialized())


def result_reduction(outs: Sequence[Any]) -> Any:
    """Per-output [sum, sum of squares, max |.|] as one float64 tensor on the
    outputs' device -- the "fused result reduction" that is exchanged."""
    import torch

    rows = []
    for o in outs:
        o64 = o.to(torch.float64)
        if o64.numel() == 0:
            rows.append(torch.zeros(3, dtype=torch.float64, device=o.device))
        else:
            rows.append(torch.stack([o64.sum(), (o64 * o64).sum(), o64.abs().max()]))
    return torch.stack(rows)


def _comm_tensor(t: Any) -> Any:
    """gloo moves host memory: stage device tensors through the CPU for it (nccl: as is)."""
    import torch.distributed as dist

    return t.cpu() if (dist.get_backend() == "gloo" and t.is_cuda) else t


def allgather_reduction(local: Any) -> Any:
    """All-gather the per-shard reductions -> tensor [world, n_outputs, 3]."""
    import torch
    import torch.distributed as dist

    if not in_group():
        return local.unsqueeze(0)
    send = _comm_tensor(local.contiguous())
    gathered = [torch.empty_like(send) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, send)
    return torch.stack(gathered).to(local.device)


def combine_reductions(gathered: Any) -> Any:
    """Global [sum, sum of squares, max |.|] per output from the gathered shards."""
    import torch

    return torch.stack([gathered[..., 0].sum(0), gathered[..., 1].sum(0),
                        gathered[..., 2].max(0).values], dim=-1)


def allgather_field(local: Any, axis: int, sizes: Sequence[int]) -> Any:
    """Optional full-field gather along the element axis (off the timed path).
    *sizes*: shard lengths of all ranks (the last shard may be longer)."""
    import torch
    import torch.distributed as dist

    if not in_group():
        return local
    moved = local.movedim(axis, 0).contiguous()
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(moved.shape[1:]), dtype=moved.dtype, device=moved.device)
    buf[: moved.shape[0]] = moved
    buf = _comm_tensor(buf)
    parts: List[Any] = [torch.empty_like(buf) for _ in sizes]
    dist.all_gather(parts, buf)
    full = torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0).to(local.device)
    return full.movedim(0, axis).contiguous()


def allgather_fields_timed(fields: Sequence[Tuple[Any, int]], sync: Any) -> dict:
    """
    The optional full-field exchange of SURVEY section 8(e): every rank receives every other rank's shard of every
    output (equal shard lengths; *fields* = ``[(local tensor, element axis)]``).  One ``all_gather_into_tensor`` per
    field on RCCL (gloo: the list form through host memory); only the collectives are between the two fences, the
    layout copy that brings the element axis to the front is not.  Every rank stages its buffers first and the ranks
    agree that all of them could (``{"skipped": ...}`` otherwise).  Returns the milliseconds, the bytes each GPU
    received from the others, that rate in GB/s (to be read against 7 xGMI links x 153 GB/s) and ``sums`` = the sum of
    every gathered field (the caller checks them against the all-gathered reductions).
    """
    import time

    import torch
    import torch.distributed as dist

    if not in_group():
        raise RuntimeError("allgather_fields_timed needs a process group")
    world = dist.get_world_size()
    # stage first, agree, then exchange: a rank that cannot allocate its receive buffers (the likely failure: world x the
    # shard per field) must not leave the others waiting inside the collective -- every rank learns of it through one
    # all-reduce of a flag and all of them skip the exchange
    staged, why = [], ""
    flag = _agree_flag(fields[0][0].device if fields else None)   # allocated BEFORE the staging attempt: a rank that has just run out
    try:                                                           # of memory must still be able to say so

        for local, axis in fields:
            moved = _comm_tensor(local.movedim(axis, 0).contiguous())
            staged.append((moved, torch.empty((world,) + tuple(moved.shape), dtype=moved.dtype, device=moved.device)))
    except (RuntimeError, MemoryError) as exc:      # torch.cuda.OutOfMemoryError is a RuntimeError
        why = f"{type(exc).__name__}: {exc}"[:200]
        staged = []          # (released before the all-reduce below)
    if not all_agree(not why, flag=flag):
        return {"skipped": "a rank could not stage its receive buffers" + (f" (this rank: {why})" if why else " (another rank)"),
                "world_size": world}
    barrier()
    sync()
    t0 = time.perf_counter()
    for moved, full in staged:
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(full, moved)
        else:
            dist.all_gather(list(full.unbind(0)), moved)
    sync()
    ms = (time.perf_counter() - t0) * 1e3
    received = sum(moved.numel() * moved.element_size() for moved, _ in staged) * (world - 1)
    return {"ms": ms, "bytes_received_per_gpu": received, "gbps_per_gpu": received / (ms * 1e-3) * 1e-9 if ms > 0 else 0.0,
            "sums": [float(full.sum().item()) for _, full in staged], "world_size": world}


def _agree_flag(device: Any = None) -> Any:
    """The one-element tensor :func:`all_agree` reduces (on the device for RCCL, on the host for gloo)."""
    import torch
    import torch.distributed as dist

    if not in_group():
        return None
    on = device if (device is not None and dist.get_backend() != "gloo") else "cpu"
    return torch.ones(1, dtype=torch.float64, device=on)


def all_agree(ok: bool, device: Any = None, flag: Any = None) -> bool:
    """True when *ok* holds on EVERY rank (one all-reduce of a flag; every rank must call it).  *flag*: a tensor from
    :func:`_agree_flag` made ahead of time -- a rank whose *ok* is "I ran out of memory" cannot be asked to allocate one."""
    import torch.distributed as dist

    if not in_group():
        return bool(ok)
    t = flag if flag is not None else _agree_flag(device)
    t.fill_(1.0 if ok else 0.0)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def barrier() -> None:
    import torch.distributed as dist

    if in_group():
        if dist.get_backend() == "nccl":
            import torch

            dist.barrier(device_ids=[torch.cuda.current_device()])   # this rank's own GPU
        else:
            dist.barrier()


def barrier_keeping_busy(keep_busy: Any = None) -> int:
    """
    A barrier that does not let this rank's GPU fall idle while it waits for the others.  An MI355X that has had nothing to do
    for more than about two milliseconds lowers its clocks, and the next milliseconds of work run 10-20 % slower (grad at E = 1e6:
    20 launches after 5 ms of idling 209 us each, after 20 ms 225, against 188: ``tools/idle_gap_probe.py``) -- ranks reach a barrier
    at different times (allocator searches of 0.1 ... 4 s), so a plain barrier in front of a short timed region would have most
    ranks time their steps on a device that has just been waiting.  Here the barrier is asynchronous and *keep_busy()* -- a few
    launches of the workload followed by a synchronize, about a millisecond -- is called until every rank has arrived.  Returns the
    number of calls.  World size 1 / no group: nothing to wait for.
    """
    import torch.distributed as dist

    if not in_group():
        return 0
    if keep_busy is None:
        barrier()
        return 0
    if dist.get_backend() == "nccl":
        import torch

        work = dist.barrier(async_op=True, device_ids=[torch.cuda.current_device()])
    else:
        work = dist.barrier(async_op=True)
    import time

    calls = 0
    deadline = time.monotonic() + 120.0     # (a backend whose handle never reports completion must not hang the job: then wait, blocking)
    while not work.is_completed() and time.monotonic() < deadline:
        keep_busy()
        calls += 1
    work.wait()
    return calls


def gather_rows(values: Sequence[float], device: Any = None) -> List[List[float]]:
    """Every rank's row of floats (equal lengths), in rank order; one all-gather of a small float64 tensor."""
    import torch
    import torch.distributed as dist

    if not in_group():
        return [[float(v) for v in values]]
    on = device if (device is not None and dist.get_backend() != "gloo") else "cpu"
    send = torch.tensor([float(v) for v in values], dtype=torch.float64, device=on)
    rows = [torch.empty_like(send) for _ in range(dist.get_world_size())]
    dist.all_gather(rows, send)
    return [[float(x) for x in r.tolist()] for r in rows]


def max_over_ranks(value: float, device: Any = None) -> float:
    import torch
    import torch.distributed as dist

    if not in_group():
        return float(value)
    on = device if (device is not None and dist.get_backend() != "gloo") else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
