"""Canonical keys and the timing-fact archive (reference: src/feinsum/sql_utils.py,
src/feinsum/canonicalization.py:1087; archive round trips as in examples/howto_autotune.py)."""

import sqlite3

import numpy as np
import pytest

import feinsum_amd as f
from feinsum_amd import sql_utils
from feinsum_amd.canonicalization import canonicalize_einsum

import dg


def test_canonical_form_ignores_names_and_operand_order():
    ref = canonicalize_einsum(dg.grad())
    same = f.einsum("ej,xre,rij->xei", f.array("w", ("N", 35)), f.array("G", (3, 3, "N")),
                    f.array("Q", (3, 35, 35)))
    renamed = f.einsum("abc,bde,ce->acd", f.array("J", (3, 3, "K")), f.array("R", (3, 35, 35)),
                       f.array("u", ("K", 35)))
    assert canonicalize_einsum(same) == ref == canonicalize_einsum(renamed)
    assert canonicalize_einsum(ref) == ref                                   # idempotent
    assert [a.name for a in ref.args[0]] == ["arg_0", "arg_1", "arg_2"]
    # different sizes, layouts or field counts are different keys
    others = [dg.grad(20), dg.grad_t(), dg.div(), dg.face_mass(4), dg.face_mass(3), dg.face_mass_ifj_fe(4),
              dg.batched_grad(2), dg.batched_div_components()]
    keys = {sql_utils._key(canonicalize_einsum(e)) for e in others + [dg.grad()]}
    assert len(keys) == len(others) + 1
    # arrays shared between rows stay shared
    fm = canonicalize_einsum(dg.face_mass(4))
    assert sorted(len({row[k].name for row in fm.args}) for k in range(3)) == [1, 1, 4]


def test_record_query_retrieve_round_trip(tmp_path):
    db = str(tmp_path / "facts.sqlite")
    dev = f.FakeCLDevice("AMD Instinct MI355X")
    grad = dg.grad()
    with pytest.raises(RuntimeError, match="timing facts table"):
        sql_utils.query(grad, dev, database=db)
    sql_utils.record_facts(grad, None, "generic", database=db, runtime_in_sec=4.0e-3, device_name=dev.name)
    sql_utils.record_facts(grad, None, "mfma", database=db, runtime_in_sec=2.0e-5, device_name=dev.name)
    sql_utils.record_facts(dg.div(), None, "mfma", database=db, runtime_in_sec=2.1e-5, device_name=dev.name)

    facts = sql_utils.query(grad, dev, database=db)
    assert sorted(q.transform_id for q in facts) == ["generic", "mfma"]
    best = max(facts, key=lambda q: q.giga_op_rate(np.float64))
    assert best.transform_id == "mfma"
    assert best.giga_op_info[np.dtype("float64")] == pytest.approx(0.798)          # archive pin: 0.798 GFLOP @1e5
    assert best.giga_op_rate("float64") == pytest.approx(0.798 / 2.0e-5)
    assert dict(sql_utils.retrieve(grad, dev, database=db)) == {"variant": "mfma"}
    only_generic = sql_utils.retrieve(grad, dev, database=db, consider_query=lambda q: q.transform_id == "generic")
    assert dict(only_generic) == {"variant": "generic"}
    # same einsum spelled differently finds the same facts; another device or einsum finds none
    spelled = f.einsum("ej,xre,rij->xei", f.array("w", ("N", 35)), f.array("G", (3, 3, "N")),
                       f.array("Q", (3, 35, 35)))
    assert len(sql_utils.query(spelled, dev, database=db)) == 2
    assert sql_utils.query(grad, f.FakeCLDevice("NVIDIA TITAN V"), database=db) == ()
    with pytest.raises(f.NoFactInDatabaseError):
        sql_utils.query(dg.face_mass(), dev, database=db, err_if_no_results=True)
    with pytest.raises(f.NoFactInDatabaseError):
        sql_utils.retrieve(grad, dev, database=db, consider_query=lambda q: False)
    with pytest.raises(ValueError):
        sql_utils.record_facts(grad, None, "no-such-variant", database=db, runtime_in_sec=1.0, device_name="x")

    timed = sql_utils.get_timed_einsums_in_db(dev, database=db)
    assert set(timed) == {canonicalize_einsum(grad), canonicalize_einsum(dg.div())}


def test_retrieve_ranks_variants_within_one_placement(tmp_path):
    """Facts taken with the timed arrays placed differently are 5-12 % apart for the SAME kernel, so `retrieve` compares
    variants within one placement: the allocator's ("split": what evaluate() allocates itself) when any fact has it, else
    the placement with the most facts."""
    db = str(tmp_path / "facts.sqlite")
    dev = f.FakeCLDevice("AMD Instinct MI355X")
    grad = dg.grad()
    rec = lambda variant, t, **params: sql_utils.record_facts(grad, None, variant, transform_params=params, database=db,   # noqa: E731
                                                              runtime_in_sec=t, device_name=dev.name)
    rec("tiled", 1.0e-4, placement="separate")
    rec("generic", 9.0e-4, placement="separate")
    assert dict(sql_utils.retrieve(grad, dev, database=db)) == {"variant": "tiled", "placement": "separate"}
    rec("mfma", 2.0e-5, placement="separate")        # a lucky torch placement ...
    rec("mfma", 2.2e-5, placement="split")           # ... does not outrank the allocator's facts
    rec("tiled", 1.1e-4, placement="split")
    assert dict(sql_utils.retrieve(grad, dev, database=db)) == {"variant": "mfma", "placement": "split"}
    only_sep = sql_utils.retrieve(grad, dev, database=db, consider_query=lambda q: q.transform_params.get("placement") == "separate")
    assert dict(only_sep) == {"variant": "mfma", "placement": "separate"}


def test_table_has_the_reference_columns(tmp_path):
    # src/feinsum/sql_utils.py:389-410
    conn = sqlite3.connect(str(tmp_path / "facts.sqlite"))
    sql_utils.record_facts(dg.face_mass(), None, "mfma", database=conn, runtime_in_sec=1e-4,
                           device_name="AMD Instinct MI355X")
    cols = [r[1] for r in conn.execute(f"PRAGMA table_info({sql_utils.TIMINGS_TABLENAME})")]
    assert cols == ["ID", "subscripts", "index_to_length", "args", "arg_to_dtype", "device_name", "transform_id",
                    "transform_params", "runtime_in_sec", "compiler_version", "giga_op_info", "timestamp"]
    row = conn.execute(f"SELECT device_name, transform_params, giga_op_info, args FROM {sql_utils.TIMINGS_TABLENAME}").fetchone()
    assert row[0] == "AMD_Instinct_MI355X" and row[1] == "{}"
    import json

    assert json.loads(row[2]) == {"float64": pytest.approx(1.704)}                      # archive pin for face-mass b=4
    assert row[3].count("arg_0") == 4


def test_shipped_archive_has_the_headline_facts():
    import os

    if not os.path.exists(sql_utils.DEFAULT_DB):
        pytest.skip("no shipped archive yet")
    dev = f.FakeCLDevice("AMD Instinct MI355X")
    tri_grad = f.einsum("xre,rij,ej->xei", f.array("J", (2, 2, "E")), f.array("R", (2, 15, 15)), f.array("u", ("E", 15)))
    for expr in (dg.grad(), dg.div(), dg.face_mass(), dg.grad(56), dg.div(56), dg.face_mass(4, Np=56, Nfp=21), tri_grad,
                 dg.mass_apply(4), dg.cross_product_batch()):
        assert dict(sql_utils.retrieve(expr, dev))["variant"] == "mfma", expr.get_subscripts()


def test_facts_of_the_reference_archives_can_be_read(golden_dir):
    """`query_reference_archive`: the einsum keys of an archive written by the reference (bliss-canonical names, here in
    the older `use_matrix` layout of data/transform_archive_v2...v5.sqlite) are re-canonicalised, so its facts are found
    for any naming and operand order of the same einsum.  Fixture: rows extracted by
    tests/golden/make_archive_sqlite_fixture.py (data the reference holds)."""
    db = str(golden_dir / "ref_archive_extract.sqlite")
    grad = sql_utils.query_reference_archive(dg.grad(), db, device_name="NVIDIA_TITAN_V")
    assert grad and {q.transform_id for q in grad} == {"xre_rij_ej_to_xei.py", "batched_xre_rij_ej_to_xei.py"}   # (b = 1 keys too)
    best = min(grad, key=lambda q: q.runtime_in_sec)
    assert best.runtime_in_sec == pytest.approx(0.3985e-3, rel=2e-3)             # 2002 GFLOP/s on the TITAN V
    assert best.giga_op_rate("float64") == pytest.approx(2002, rel=2e-3)
    assert best.transform_params["n_e_per_wg"] == 21 and best.transform_params["nwork_items_per_e"] == 12
    # the same einsum with other names and another operand order
    other = f.einsum("pq,mnp,nkq->mpk", f.array("field", ("Nel", 35)), f.array("jac", (3, 3, "Nel")),
                     f.array("diff", (3, 35, 35)))
    renamed = sql_utils.query_reference_archive(other, db, device_name="NVIDIA_TITAN_V")
    assert sorted(q.runtime_in_sec for q in renamed) == sorted(q.runtime_in_sec for q in grad)
    div = sql_utils.query_reference_archive(dg.div(), db)
    assert min(q.runtime_in_sec for q in div) == pytest.approx(0.4159e-3, rel=2e-3)
    lift = sql_utils.query_reference_archive(dg.face_mass_ifj_fe(4), db)
    assert min(q.runtime_in_sec for q in lift) == pytest.approx(0.7795e-3, rel=2e-3)
    assert all(q.giga_op_info[np.dtype("float64")] == pytest.approx(1.704) for q in lift)
    # an einsum the archive does not hold, and a device it has no facts for
    assert sql_utils.query_reference_archive(dg.grad(20), db, device_name="AMD_Instinct_MI355X") == ()
    assert sql_utils.query_reference_archive(f.einsum("ij,jk->ik", f.array("A", (7, 7)), f.array("B", (7, 7))), db) == ()
