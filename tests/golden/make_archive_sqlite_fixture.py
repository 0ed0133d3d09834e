#!/usr/bin/env python
"""
A small extract of the reference's timing-fact archives, as sqlite, for tests/test_sql_utils.py.

Runs in the build container only (``/root/reference`` does not travel to the GPU box):

    python tests/golden/make_archive_sqlite_fixture.py

Copies ROWS (data the reference holds; no reference source) from ``data/transform_archive_v5.sqlite`` -- the five best
and the worst fact of every p = 4 key with three operands (grad / div / face-mass and their siblings), and the facts of
two streaming keys -- and the first 20 rows of one per-device table of ``data/transform_archive_v2.sqlite`` into
``tests/golden/ref_archive_extract.sqlite``, keeping the tables' own layouts (v5: FEINSUM_TIMING_FACTS with the older
``use_matrix`` / ``value_to_dtype`` columns; v2: one table per device).
"""

import json
import sqlite3
from pathlib import Path

REF = Path("/root/reference/data")
OUT = Path(__file__).resolve().parent / "ref_archive_extract.sqlite"


def main() -> None:
    OUT.unlink(missing_ok=True)
    out = sqlite3.connect(OUT)
    src = sqlite3.connect(f"file:{REF / 'transform_archive_v5.sqlite'}?mode=ro", uri=True)
    (ddl,) = src.execute("select sql from sqlite_master where name = 'FEINSUM_TIMING_FACTS'").fetchone()
    out.execute(ddl)
    cols = [r[1] for r in src.execute("pragma table_info(FEINSUM_TIMING_FACTS)")]
    keys = src.execute("select distinct subscripts, index_to_length, use_matrix, value_to_dtype from FEINSUM_TIMING_FACTS").fetchall()
    n = 0
    for subs, i2l, um, v2d in keys:
        sizes = sorted(json.loads(i2l).values())
        three = subs.split("->")[0].count(",") == 2
        if not ((three and 35 in sizes) or subs in ("ab,b->a", "ab->a")):
            continue
        rows = src.execute(f"select {', '.join(cols)} from FEINSUM_TIMING_FACTS where subscripts = ? and index_to_length = ? "
                           "and use_matrix = ? and value_to_dtype = ? order by runtime_in_sec", (subs, i2l, um, v2d)).fetchall()
        for row in rows[:5] + rows[-1:]:
            out.execute(f"insert into FEINSUM_TIMING_FACTS ({', '.join(c for c in cols if c != 'ID')}) values "
                        f"({', '.join('?' for c in cols if c != 'ID')})", [v for c, v in zip(cols, row) if c != "ID"])
            n += 1
    src.close()
    src = sqlite3.connect(f"file:{REF / 'transform_archive_v2.sqlite'}?mode=ro", uri=True)
    (table, ddl), = src.execute("select name, sql from sqlite_master where type = 'table' and name not like 'sqlite_%' limit 1").fetchall()
    out.execute(ddl)
    cols = [r[1] for r in src.execute(f"pragma table_info({table})")]
    for row in src.execute(f"select {', '.join(cols)} from {table} order by ID limit 20"):
        out.execute(f"insert into {table} ({', '.join(c for c in cols if c != 'ID')}) values ({', '.join('?' for c in cols if c != 'ID')})",
                    [v for c, v in zip(cols, row) if c != "ID"])
        n += 1
    src.close()
    out.commit()
    out.execute("vacuum")
    out.close()
    print(f"{n} facts -> {OUT} ({OUT.stat().st_size} bytes)")


if __name__ == "__main__":
    main()
