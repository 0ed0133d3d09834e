"""N > 1 path on CPU: world_size-2 gloo processes shard the elements, evaluate
their shard (the oracle stands in for the GPU kernels here), all-gather the
result reductions and agree with the unsharded evaluation."""

import os
import socket

import numpy as np
import pytest

from feinsum_amd import parallel

import dg


def test_shard_bounds_cover_and_align():
    for E in (0, 1, 15, 16, 17, 100, 1000, 10**6, 10**6 + 7):
        for world in (1, 2, 3, 4, 8):
            spans = [parallel.shard_bounds(E, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == E
            for (s0, e0), (s1, e1) in zip(spans, spans[1:]):
                assert e0 == s1
            assert all(s % parallel.TILE == 0 for s, _ in spans)
            sizes = [e - s for s, e in spans]
            assert all(sz == sizes[0] for sz in sizes[:-1]) and sizes[-1] >= sizes[0]
    assert parallel.shard_bounds(8_000_000, 8, 3) == (3_000_000, 4_000_000)
    with pytest.raises(ValueError):
        parallel.shard_bounds(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, E, tmpdir):
    import torch
    import torch.distributed as dist

    from oracle import np_oracle
    from feinsum_amd.measure import generate_host_input_arrays

    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    info = parallel.init_distributed("gloo")
    assert (info.rank, info.world_size, info.backend) == (rank, world, "gloo")
    expr = dg.grad()
    host = generate_host_input_arrays(expr, E)           # same seed on every rank
    mine = parallel.shard_host_arrays(expr, host, world, rank)
    s, e = parallel.shard_bounds(E, world, rank)
    assert mine["u"].shape == (e - s, 35) and mine["J"].shape == (3, 3, e - s)
    assert mine["R"].shape == (3, 35, 35)
    out = np_oracle.reference_outputs(expr.get_subscripts(), [[mine[a.name] for a in expr.args[0]]])[0]
    local = parallel.result_reduction([torch.from_numpy(out)])
    gathered = parallel.allgather_reduction(local)
    assert tuple(gathered.shape) == (world, 1, 3)
    total = parallel.combine_reductions(gathered)
    sizes = [b - a for a, b in (parallel.shard_bounds(E, world, r) for r in range(world))]
    full = parallel.allgather_field(torch.from_numpy(out), 1, sizes)
    parallel.barrier()
    t = parallel.max_over_ranks(float(rank))
    assert t == world - 1
    if rank == 0:
        np.save(os.path.join(tmpdir, "total.npy"), total.numpy())
        np.save(os.path.join(tmpdir, "full.npy"), full.numpy())
    dist.destroy_process_group()


def test_two_rank_gloo_sharded_grad(tmp_path):
    import torch.multiprocessing as mp

    from oracle import np_oracle
    from feinsum_amd.measure import generate_host_input_arrays

    E, world = 83, 2   # 83 = 2 x 32 + 19: ragged last shard
    mp.spawn(_worker, args=(world, _free_port(), E, str(tmp_path)), nprocs=world, join=True)
    expr = dg.grad()
    host = generate_host_input_arrays(expr, E)
    ref = np_oracle.reference_outputs(expr.get_subscripts(), [[host[a.name] for a in expr.args[0]]])[0]
    total = np.load(tmp_path / "total.npy")
    full = np.load(tmp_path / "full.npy")
    np.testing.assert_array_equal(full, ref)             # shards are bit-identical slices
    np.testing.assert_allclose(total[0], [ref.sum(), (ref * ref).sum(), np.abs(ref).max()], rtol=1e-13)


def test_group_of_one_runs_the_collectives():
    """FEINSUM_DIST_FORCE=1: a single process joins a group and every exchange goes through the backend
    (how a one-GPU box rehearses the RCCL calls; here with gloo)."""
    import subprocess
    import sys

    code = (
        "import os, torch\n"
        "from feinsum_amd import parallel\n"
        "info = parallel.init_distributed()\n"
        "assert parallel.in_group() and info.world_size == 1 and info.backend == 'gloo'\n"
        "local = parallel.result_reduction([torch.arange(6, dtype=torch.float64).reshape(2, 3)])\n"
        "g = parallel.allgather_reduction(local)\n"
        "assert g.shape == (1, 1, 3) and torch.equal(parallel.combine_reductions(g), local)\n"
        "parallel.barrier()\n"
        "assert parallel.max_over_ranks(2.5) == 2.5\n"
        "f = parallel.allgather_field(torch.ones(4, 3, dtype=torch.float64), 0, [4])\n"
        "assert f.shape == (4, 3)\n"
        "import torch.distributed as dist; dist.destroy_process_group()\n"
        "print('ok')\n")
    env = dict(os.environ, FEINSUM_DIST_FORCE="1", FEINSUM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29641", WORLD_SIZE="1", RANK="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
