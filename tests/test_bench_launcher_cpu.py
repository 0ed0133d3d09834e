"""`python bench.py --gpus N` started without a launcher becomes the parent of N rank processes:
the decision, the environment each rank receives, the relay of rank 0's line, and the exit code when a rank fails.
The ranks here are a stub script (no GPU in this container); the real ranks run on the GPU box
(tests/test_gpu_bench_line.py::test_bench_spawns_its_own_ranks)."""

import io
import json
import os
import sys
import textwrap
import time
from pathlib import Path

import bench

ROOT = Path(__file__).resolve().parents[1]


def test_decision_to_spawn():
    assert bench.needs_own_ranks(8, {})
    assert bench.needs_own_ranks(2, {"PATH": "/bin"})
    assert not bench.needs_own_ranks(1, {})                                    # the driver's one-GPU line
    assert not bench.needs_own_ranks(8, {"RANK": "3", "WORLD_SIZE": "8"})     # under torch.distributed.run
    assert not bench.needs_own_ranks(8, {"WORLD_SIZE": "8"})


def _stub(tmp_path, body: str) -> Path:
    script = tmp_path / "rank_stub.py"
    script.write_text(textwrap.dedent(body))
    return script


def test_every_rank_gets_its_environment_and_rank0_is_relayed(tmp_path):
    script = _stub(tmp_path, """
        import json, os, sys
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY",
                "FEINSUM_BENCH_LAUNCHER")
        rec = {k: os.environ.get(k) for k in keys}
        rec["argv"] = sys.argv[1:]
        open(os.path.join(os.environ["STUB_DIR"], "rank%s.json" % os.environ["RANK"]), "w").write(json.dumps(rec))
        print(json.dumps({"rank": int(os.environ["RANK"]), "n_gpus": int(os.environ["WORLD_SIZE"])}), flush=True)
    """)
    out, err = io.StringIO(), io.StringIO()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["STUB_DIR"] = str(tmp_path)
    rc = bench.spawn_ranks(4, ["--gpus", "4", "--steps", "3"], script=script, environ=env, out=out, err=err, timeout_s=60)
    assert rc == 0, err.getvalue()
    recs = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(4)]
    assert [r["RANK"] for r in recs] == ["0", "1", "2", "3"] and [r["LOCAL_RANK"] for r in recs] == ["0", "1", "2", "3"]
    assert {r["WORLD_SIZE"] for r in recs} == {"4"} and {r["MASTER_ADDR"] for r in recs} == {"127.0.0.1"}
    assert len({r["MASTER_PORT"] for r in recs}) == 1 and int(recs[0]["MASTER_PORT"]) > 0
    assert {r["HSA_ENABLE_IPC_MODE_LEGACY"] for r in recs} == {"0"}
    assert all(r["argv"] == ["--gpus", "4", "--steps", "3"] for r in recs)
    assert all("self-spawned" in r["FEINSUM_BENCH_LAUNCHER"] for r in recs)
    lines = [ln for ln in out.getvalue().splitlines() if ln.startswith("{")]
    assert lines == [json.dumps({"rank": 0, "n_gpus": 4})]                    # ONE line on stdout: rank 0's
    assert err.getvalue().count('"rank"') == 3                                # the other ranks' output goes to stderr


def test_a_failing_rank_fails_the_run_and_stops_the_others(tmp_path):
    script = _stub(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(120)      # a rank stuck in a collective its peer never joins
    """)
    out, err = io.StringIO(), io.StringIO()
    t0 = time.monotonic()
    rc = bench.spawn_ranks(3, [], script=script, out=out, err=err, timeout_s=100)
    assert rc != 0 and time.monotonic() - t0 < 60
    assert "rank 1 exited with 7" in err.getvalue()


def test_a_rank_that_fails_behind_the_line_cannot_take_the_line_with_it(tmp_path):
    """The optional full-field gather runs AFTER rank 0 has printed the JSON line (bench.py --gather-fields on): one rank
    raising inside it while the others wait in the collective still leaves the relayed line, and the parent ends the
    waiting ranks and exits 1 within seconds -- deterministically."""
    script = _stub(tmp_path, """
        import json, os, sys, time
        rank = int(os.environ["RANK"])
        if rank == 0:
            print(json.dumps({"metric": "stub", "value": 1.0, "n_gpus": int(os.environ["WORLD_SIZE"])}), flush=True)
        time.sleep(0.3)                      # every rank is behind the line now
        if rank == 2:
            raise RuntimeError("out of memory while staging the receive buffers")
        time.sleep(120)                      # the others wait inside the collective
    """)
    out, err = io.StringIO(), io.StringIO()
    t0 = time.monotonic()
    rc = bench.spawn_ranks(4, [], script=script, out=out, err=err, timeout_s=100)
    assert rc == 1 and time.monotonic() - t0 < 30
    lines = [ln for ln in out.getvalue().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 4
    assert "rank 2 exited with 1" in err.getvalue()


def test_timeout_stops_the_ranks(tmp_path):
    script = _stub(tmp_path, "import time; time.sleep(120)")
    err = io.StringIO()
    t0 = time.monotonic()
    rc = bench.spawn_ranks(2, [], script=script, out=io.StringIO(), err=err, timeout_s=1.0)
    assert rc != 0 and time.monotonic() - t0 < 60 and "no result after" in err.getvalue()


def test_parent_touches_no_gpu_module():
    """The parent path returns before torch is imported: `bench.py --gpus 2` in a process where importing torch is
    forbidden still spawns (the stub ranks exit at once because the real script needs a device)."""
    import subprocess

    code = ("import sys, builtins; real = builtins.__import__\n"
            "def guard(name, *a, **k):\n"
            "    assert not name.startswith('torch'), 'the parent imported ' + name\n"
            "    return real(name, *a, **k)\n"
            "builtins.__import__ = guard\n"
            "import bench\n"
            "bench.spawn_ranks = lambda n, argv, **kw: print('SPAWN', n, argv) or 0\n"
            "sys.argv = ['bench.py', '--gpus', '2', '--steps', '3']\n"
            "bench.main()\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")}
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "SPAWN 2 ['--gpus', '2', '--steps', '3']" in res.stdout
