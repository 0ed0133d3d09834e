#!/usr/bin/env python
"""
Extract the reference-held operation counts into a fixture.

Runs in the build container only (``/root/reference`` does not travel to the GPU box):

    python tests/golden/make_archive_fixture.py

Reads the timing-fact archives the reference ships, ``data/transform_archive_v2.sqlite`` ...
``_v5.sqlite`` (table layout: ``src/feinsum/sql_utils.py:389-415``; v2 keeps one table per
device, v3+ the single ``FEINSUM_TIMING_FACTS`` table), and writes one record per distinct einsum
key to ``tests/golden/ref_archive_opcounts.json``:

    archive, subscripts, index_to_length, use_matrix (the argument-name matrix: one row per
    batch member, one list of value names per array access), value_to_dtype, giga_op_info
    (the reference's algorithmic GOp count, ``src/feinsum/measure.py:278-331`` evaluated at the
    long-dimension length the fact was recorded at), the number of facts, and the best recorded
    TITAN V run time with its transform.

These are DATA the reference holds -- the only numerical fixtures it has besides the two integers
of ``test/test_loopy_utils.py:270-271`` -- not reference source.  ``tests/test_archive_opcounts.py``
checks the build's own counter (``feinsum_amd.count_ops``) against every record.
"""

from __future__ import annotations

import json
import sqlite3
from pathlib import Path

REF_DATA = Path("/root/reference/data")
OUT = Path(__file__).resolve().parent / "ref_archive_opcounts.json"


def _facts(version: int):
    path = REF_DATA / f"transform_archive_v{version}.sqlite"
    conn = sqlite3.connect(f"file:{path}?mode=ro", uri=True)
    tables = [name for (name,) in conn.execute("select name from sqlite_master where type='table'")
              if not name.startswith("sqlite_")]
    for table in tables:
        cols = [row[1] for row in conn.execute(f"pragma table_info({table})")]
        device_expr = "device_name" if "device_name" in cols else f"'{table}'"
        yield from conn.execute(
            f"select subscripts, index_to_length, use_matrix, value_to_dtype, giga_op_info, "
            f"runtime_in_sec, transform_id, {device_expr} from {table} order by ID")
    conn.close()


def main() -> None:
    records = []
    for version in (2, 3, 4, 5):
        by_key: dict = {}
        for subs, i2l, um, v2d, gops, runtime, transform_id, device in _facts(version):
            rec = by_key.setdefault((subs, i2l, um, v2d), {
                "archive": f"transform_archive_v{version}.sqlite",
                "subscripts": subs,
                "index_to_length": json.loads(i2l),
                "use_matrix": json.loads(um),
                "value_to_dtype": json.loads(v2d),
                "giga_op_info": json.loads(gops),
                "device": device,
                "n_facts": 0,
                "best_runtime_in_sec": None,
                "best_transform_id": None,
            })
            if json.loads(gops) != rec["giga_op_info"]:
                raise SystemExit(f"inconsistent giga_op_info for key {subs} {i2l}")
            rec["n_facts"] += 1
            if rec["best_runtime_in_sec"] is None or runtime < rec["best_runtime_in_sec"]:
                rec["best_runtime_in_sec"], rec["best_transform_id"] = runtime, transform_id
        records += list(by_key.values())
    OUT.write_text(json.dumps({"source": "kaushikcfd/feinsum data/transform_archive_v{2,3,4,5}.sqlite",
                               "records": records}, indent=1, sort_keys=True) + "\n")
    print(f"{len(records)} einsum keys, {sum(r['n_facts'] for r in records)} facts -> {OUT}")


if __name__ == "__main__":
    main()
