"""Record timing facts for the kernel variants of an einsum and ask the archive for the best one --
the loop of the reference's ``examples/howto_autotune.py`` (``f.autotune`` records facts with
``record_facts``; users then call ``f.query`` / ``f.retrieve``), with kernel variants in place of
points of a loopy transform space.

    python examples/howto_archive.py [database.sqlite]
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import feinsum_amd as f  # noqa: E402


def main():
    db = sys.argv[1] if len(sys.argv) > 1 else "/tmp/feinsum_facts.sqlite"
    expr = f.einsum("xre,rij,ej->xei", f.array("J", (3, 3, "Nel")), f.array("R", (3, 35, 35)), f.array("u", ("Nel", 35)))
    q = f.DeviceQueue(0)
    for variant in ("generic", "tiled", "mfma"):
        f.record_facts(expr, q, variant, database=db, long_dim_length=100_000)
    for fact in f.query(expr, q.device, database=db):
        print(f"{fact.transform_id:8s} {fact.runtime_in_sec * 1e6:9.1f} us  {fact.giga_op_rate(np.float64):9.0f} GFLOP/s")
    best = f.retrieve(expr, q.device, database=db)
    print("best variant:", dict(best))
    print("timeit with it:", f.timeit(expr, cq=q, transform=best, long_dim_length=100_000), "s")


if __name__ == "__main__":
    main()
