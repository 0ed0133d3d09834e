"""
bench.py end to end on the GPU box, as the driver starts it: the JSON line carries the contract's
fields, and the sharded path's collectives run through RCCL itself (a one-GPU box joins a group of
one: FEINSUM_DIST_FORCE=1), so the N > 1 launch cannot fail on the backend's calls.
"""

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
SMALL = ["--steps", "3", "--warmup", "1", "--setup-launches", "5", "--setup-seconds", "0", "--elems-per-gpu", "20000", "--no-cpu-baseline",
         "--no-protocol", "--placement", "separate"]
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "setup_launches"}


def _json_line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["grad", "pipeline"])
def test_bench_line_single_process(workload):
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--workload", workload] + SMALL,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _json_line(out.stdout)
    assert REQUIRED <= set(line), sorted(REQUIRED - set(line))
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["dtype"] == "f64" and line["result_finite"]
    assert line["dist_backend"] is None and line["value"] > 0
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    # 20 000 elements: one round, static walk -- grad stores write-through, the fused pipeline does not (fe_common.h)
    assert line["stores"]["policy"].startswith("write-through" if workload == "grad" else "non-temporal"), line["stores"]


@pytest.mark.gpu
def test_bench_collectives_through_rccl_in_a_group_of_one():
    env = dict(os.environ, FEINSUM_DIST_FORCE="1", MASTER_ADDR="127.0.0.1")
    env.pop("FEINSUM_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29631", str(ROOT / "bench.py"), "--gpus", "1", "--workload", "pipeline", "--gather-fields", "on"] + SMALL
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _json_line(out.stdout)
    assert line["dist_backend"] == "nccl" and line["n_gpus"] == 1 and line["result_finite"] and line["value"] > 0
    assert line["result_allgather_ms"] >= 0.0
    # the full-field exchange runs BEHIND the line and is reported on stderr: six output fields through
    # all_gather_into_tensor (a group of one receives nothing)
    assert "after this line" in line["field_allgather"]
    reports = [ln for ln in out.stderr.splitlines() if ln.startswith("field_allgather {")]
    assert len(reports) == 1, out.stderr[-2000:]
    g = json.loads(reports[0][len("field_allgather "):])
    assert g["world_size"] == 1 and g["bytes_received_per_gpu"] == 0 and g["matches_reduction"], g


@pytest.mark.gpu
@pytest.mark.parametrize("workload,placement", [("grad", "split"), ("facemass", "split")])
def test_bench_default_placement(workload, placement):
    """The DEFAULT bench path (outputs from the split allocator), which `--placement separate` in the tests above never
    runs."""
    small = [a for a in SMALL if a not in ("--placement", "separate", "--no-protocol")]
    small[small.index("20000")] = "1000000"     # (outputs large enough to split)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--workload", workload, "--placement", placement] + small
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _json_line(out.stdout)
    assert line["placement"]["mode"] == placement and line["result_finite"] and line["value"] > 0
    assert line["kernel_ms_separate_allocations"] > 0            # the A/B against torch allocations stays in the line
    assert line["walk"]["kernel_ms_static_walk"] > 0 and line["walk"]["mode"].startswith("tickets")   # ... and the A/B of the walks
    # (round 5: what the line says about walk, loads and stores is the launcher's own record -- fe_last_launch_info)
    assert line["walk"]["tiles"] == 62500 * (1 if workload == "grad" else 1) and 0 < line["walk"]["static_tiles"] < line["walk"]["tiles"]
    assert line["loads"]["streamed_operand"].startswith("non-temporal") and line["stores"]["policy"] == "non-temporal"
    assert line["per_rank"][0]["rank"] == 0 and line["per_rank"][0]["elements"] == 1_000_000 and line["per_rank"][0]["kernel_ms"] > 0
    assert line["ms_per_step_barrier_inclusive"] >= line["ms_per_step"] and line["placement"]["degraded"] in (True, False)
    assert line["config"]["variant"] == "auto"                   # no silent switch of the kernel variant
    if placement == "split":
        rep = line["placement"]
        assert len(rep["output_pieces_by_class"]) == (1 if workload == "grad" else 4)
        if rep["pool"]["classes"] >= 2 and not rep["pool"]["unsplit_arrays"]:      # alternating pieces: two classes, equal shares
            for counts in rep["output_pieces_by_class"]:
                used = sorted(c for c in counts if c)
                assert len(used) == 2 and used[1] - used[0] <= 1, counts
        assert rep["output_bytes"] <= rep["output_mapped_bytes"] < rep["output_bytes"] + 4 * (2 << 20)
        assert rep["allocator_ms"] < 5000 and "scan_positions" not in rep


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent starts two ranks (here sharing the one GPU through gloo)
    and relays rank 0's line; the parent logic itself is covered on CPU in test_bench_launcher_cpu.py."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["FEINSUM_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "grad"] + SMALL,
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _json_line(out.stdout)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["dist_backend"] == "gloo"
    assert "self-spawned" in line["launcher"] and line["result_finite"] and line["value"] > 0
    assert line["config"]["elements_total"] == 2 * line["config"]["elements_per_gpu"]
    assert [r["rank"] for r in line["per_rank"]] == [0, 1] and all(r["kernel_ms"] > 0 and r["placement_mode"] == "separate" for r in line["per_rank"])
