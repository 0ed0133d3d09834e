"""bench.py's rank logic with world_size 2 on CPU (gloo): element split, barrier-bracketed timed
region with MAX over ranks, the result-reduction exchange, and the JSON line built from them.
The per-shard evaluator here is the oracle (test infrastructure); on the GPU box it is the HIP path
-- the code under test is everything around it."""

import argparse
import json
import os
import socket
import time

import numpy as np
import pytest

import dg


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, elems_total, elems_per_gpu, tmpdir):
    import torch
    import torch.distributed as dist

    import bench
    import feinsum_amd as f
    from feinsum_amd import parallel
    from feinsum_amd.measure import generate_host_input_arrays
    from oracle import np_oracle

    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    info = parallel.init_distributed("gloo")
    args = argparse.Namespace(elems_total=elems_total, elems_per_gpu=elems_per_gpu)
    E = bench.rank_elements(args, info)
    expr = dg.grad()
    if elems_total:      # one global batch: this rank's block of the same arrays
        host = parallel.shard_host_arrays(expr, generate_host_input_arrays(expr, elems_total), world, rank)
    else:                # weak scaling: every rank draws its own batch
        host = generate_host_input_arrays(expr, E, np_seed=1000 * rank)
    assert host["u"].shape[0] == E
    out = np_oracle.reference_outputs(expr.get_subscripts(), [[host[a.name] for a in expr.args[0]]])[0]

    # a rank-dependent "kernel": rank r's batch of n steps takes 20 (r + 1) ms per step
    def step_batch(n):
        time.sleep(0.02 * (rank + 1) * n)
        return 0.01 * (rank + 1) * n

    # rank 1 reaches the opening barrier 0.3 s late: rank 0 keeps "its device busy" meanwhile (parallel.barrier_keeping_busy)
    calls = []
    if rank == 1:
        time.sleep(0.3)
    timed = bench.timed_region(step_batch, 3, sync=lambda: None, keep_busy=lambda: (calls.append(1), time.sleep(0.002)))
    assert timed["opening_barrier_busy_calls"] == len(calls)
    assert (len(calls) >= 30) if rank == 0 else (len(calls) <= 20), (rank, len(calls))
    wall_s, kernel_s = timed["wall"], timed["kernel"]
    # the closing barrier is outside `wall`: rank 0 is done after 3 x 20 ms and only then waits for rank 1
    assert timed["local_wall"] == pytest.approx(0.06 * (rank + 1), abs=0.03) and timed["local_kernel"] == pytest.approx(0.03 * (rank + 1))
    assert timed["wall_barrier"] >= timed["wall"] >= timed["local_wall"]
    per_rank = bench.gather_rank_reports({"elements": E, "kernel_ms": timed["local_kernel"] / 3 * 1e3, "wall_ms": timed["local_wall"] / 3 * 1e3,
                                          "placement_mode": "split", "unsplit_arrays": rank, "allocator_ms": 10.0 * rank,
                                          "busy_calls": timed["opening_barrier_busy_calls"]})
    total, red_ms, gather_ms = bench.exchange_results([torch.from_numpy(out)], sync=lambda: None)
    flops = float(f.count_ops(expr, long_dim_length=E))
    t = torch.tensor([flops], dtype=torch.float64)
    dist.all_reduce(t)
    if not elems_total:  # equal shards: the optional full-field exchange
        g = parallel.allgather_fields_timed([(torch.from_numpy(out), 1)], sync=lambda: None)
        assert g["world_size"] == world and g["bytes_received_per_gpu"] == out.nbytes * (world - 1)
        assert abs(g["sums"][0] - float(total[0, 0])) <= 1e-9 * abs(float(total[0, 0]))
    if rank == 0:
        line = bench.compose_line(workload="grad", n_gpus=world, steps=3, warmup=0, wall_s=wall_s, kernel_s=kernel_s,
                                  flops_step_all=float(t.item()) * 1e6, flops_step_rank0=flops * 1e6,   # (x 1e6: the line rounds to 0.1 GFLOP/s)
                                  bytes_step_rank0=8.0 * (149 * E + 3675) * 1e6, elems_rank0=E, elems_total=elems_total,
                                  variant="auto", device_name="cpu rehearsal", entry_points=("fe_grad",),
                                  extra={"result_reduction_ms": red_ms, "result_allgather_ms": gather_ms})
        json.dump({"line": line, "total": total.tolist(), "wall_s": wall_s, "kernel_s": kernel_s, "per_rank": per_rank},
                  open(os.path.join(tmpdir, "rank0.json"), "w"))
    dist.destroy_process_group()


@pytest.mark.parametrize("elems_total,elems_per_gpu", [(83, 0), (0, 40)])
def test_two_rank_bench_logic(tmp_path, elems_total, elems_per_gpu):
    import torch.multiprocessing as mp

    from feinsum_amd.measure import generate_host_input_arrays
    from oracle import np_oracle

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), elems_total, elems_per_gpu, str(tmp_path)), nprocs=world, join=True)
    rec = json.load(open(tmp_path / "rank0.json"))
    line = rec["line"]
    # MAX over ranks: rank 1 sleeps 3 x 40 ms, rank 0 only 3 x 20 ms
    assert 0.12 <= rec["wall_s"] < 0.5 and rec["kernel_s"] == pytest.approx(0.06)
    expr = dg.grad()
    if elems_total:
        host = generate_host_input_arrays(expr, elems_total)
        ref = np_oracle.reference_outputs(expr.get_subscripts(), [[host[a.name] for a in expr.args[0]]])[0]
        n_all, e0 = elems_total, 32          # 83 = 32 + 51: tile-aligned block, ragged tail on the last rank
        assert line["scaling"] == "strong"
    else:
        refs = []
        for r in range(world):
            host = generate_host_input_arrays(expr, elems_per_gpu, np_seed=1000 * r)
            refs.append(np_oracle.reference_outputs(expr.get_subscripts(), [[host[a.name] for a in expr.args[0]]])[0])
        ref = np.concatenate(refs, axis=1)
        n_all, e0 = world * elems_per_gpu, elems_per_gpu
        assert line["scaling"] == "weak"
    np.testing.assert_allclose(rec["total"][0], [ref.sum(), (ref * ref).sum(), np.abs(ref).max()], rtol=1e-13)
    # whole-job value = all ranks' flops / max-over-ranks step time
    assert line["value"] == pytest.approx(7980 * n_all * 1e6 / (rec["wall_s"] / 3) * 1e-9, rel=1e-3)
    assert line["n_gpus"] == 2 and line["config"]["elements_per_gpu"] == e0
    assert line["config"]["elements_total"] == n_all
    assert line["kernel_ms"] == pytest.approx(20.0)
    assert line["roofline"]["achieved"] == pytest.approx(8.0 * (149 * e0 + 3675) * 1e6 / 0.02 * 1e-9, rel=1e-3)
    assert {"result_reduction_ms", "result_allgather_ms", "setup_launches", "ms_per_step"} <= set(line)
    assert line["roofline"]["traffic"] is None
    # every rank's own figures reach rank 0: a slow rank (or one whose allocator search failed) is attributable
    pr = rec["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and [r["unsplit_arrays"] for r in pr] == [0, 1]
    assert pr[0]["opening_barrier_busy_calls"] >= 30 and pr[1]["opening_barrier_busy_calls"] <= 20      # rank 0 waited for rank 1, busy
    assert pr[1]["kernel_ms"] == pytest.approx(20.0) and pr[0]["kernel_ms"] == pytest.approx(10.0)
    assert pr[1]["wall_ms"] > pr[0]["wall_ms"] and pr[0]["placement_mode"] == "split"


def test_committed_counters_are_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    import bench

    sha = bench.kernel_source_sha()
    assert len(sha) == 16 and sha == bench.kernel_source_sha()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: sha)
    rec = {"E": 1000, "source_sha": sha, "hbm_bytes_per_launch": 123.0, "kernel": "k"}
    (prof / "traffic_grad_E1000.json").write_text(json.dumps(rec))        # (E = 1e6: traffic_grad.json; other sizes carry theirs)
    assert bench.committed_counters("grad", 1000)[0]["hbm_bytes_per_launch"] == 123.0
    assert bench.committed_counters("grad", 2000)[0] is None
    (prof / "traffic_grad.json").write_text(json.dumps(rec))
    assert bench.committed_counters("grad", 1_000_000) == (None, "committed PMC profile is for E=1000")
    (prof / "traffic_grad_E1000.json").write_text(json.dumps(dict(rec, source_sha="0" * 16)))
    got, note = bench.committed_counters("grad", 1000)
    assert got is None and "kernel sources" in note
    assert bench.committed_counters("div", 1000)[0] is None


def test_kernel_source_hash_sees_what_the_compiler_sees(tmp_path, monkeypatch):
    """A committed PMC profile is tied to the kernel sources by a hash that ignores comments and white space: round 3's last
    commit edited one comment and the driver's line lost `roofline.traffic` / `mfma_util` to it."""
    import bench

    text = """// header comment
    #include <x.h>   /* why */
    __global__ void k(int* p) {   // a kernel
        const char* s = "// not a comment";  p[0] = '/' + s[0];
    }
    """
    assert bench.strip_c_comments(text) == \
        '#include <x.h> __global__ void k(int* p) { const char* s = "// not a comment"; p[0] = \'/\' + s[0]; }'
    src = tmp_path / "feinsum_amd" / "csrc"
    src.mkdir(parents=True)
    (tmp_path / "include").mkdir()
    (tmp_path / "include" / "feinsum_hip.h").write_text("int fe_version(void); /* v1 */\n")
    (src / "k.hip").write_text(text)
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    sha = bench.kernel_source_sha()
    (src / "k.hip").write_text(text.replace("// a kernel", "// the kernel, described at greater length\n"))
    (tmp_path / "include" / "feinsum_hip.h").write_text("int fe_version(void);\n\n/* v2 */\n")
    assert bench.kernel_source_sha() == sha                      # prose and layout do not move it
    (src / "k.hip").write_text(text.replace("p[0] =", "p[1] ="))
    assert bench.kernel_source_sha() != sha                      # code does


def _gather_worker(rank, world, port, tmpdir):
    import torch

    from feinsum_amd import parallel

    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init_distributed("gloo")
    assert parallel.all_agree(True) is True
    assert parallel.all_agree(rank != 1) is False                # one rank's "no" reaches every rank
    field = torch.arange(12, dtype=torch.float64).reshape(3, 4) + 100 * rank
    good = parallel.allgather_fields_timed([(field, 0)], sync=lambda: None)
    assert good["world_size"] == world and len(good["sums"]) == 1

    class Unstageable:                   # rank 1 cannot allocate its receive buffer
        device = field.device

        def movedim(self, *a):
            raise RuntimeError("HIP out of memory (rehearsal)")

    res = parallel.allgather_fields_timed([(Unstageable() if rank == 1 else field, 0)], sync=lambda: None)
    with open(os.path.join(tmpdir, f"gather{rank}.json"), "w") as fh:
        json.dump(res, fh)
    import torch.distributed as dist

    dist.destroy_process_group()


def test_no_rank_enters_the_field_gather_alone(tmp_path):
    """parallel.allgather_fields_timed: a rank that cannot stage its receive buffers tells the others through one all-reduce
    of a flag and EVERY rank skips the collective (nobody waits inside it for a peer that never comes)."""
    import torch.multiprocessing as mp

    mp.spawn(_gather_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (json.load(open(tmp_path / f"gather{r}.json")) for r in (0, 1))
    assert "another rank" in r0["skipped"] and "out of memory" in r1["skipped"]
