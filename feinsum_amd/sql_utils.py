"""
Timing facts in the reference's archive format, for the HIP kernel variants.

API mirror of ``src/feinsum/sql_utils.py``: :func:`record_facts` (``:413-509``),
:func:`query` (``:160-246``), :func:`retrieve` (``:249-294``),
:func:`get_timed_einsums_in_db` (``:342-386``), :class:`QueryInfo` (``:133-157``).
The sqlite table has the reference's name and columns (``:389-410``) and the same
JSON encodings of every column (``:54-130``), so the reference's own tooling can
read a database written here; what differs is what the columns *mean*:

``transform_id``      the reference stores the file name of a loopy transform
                      (``xre_rij_ej_to_xei.py``); here it is the name of a kernel
                      variant of ``libfeinsum_hip.so`` (``"mfma"``, ``"generic"``).
``transform_params``  what else parametrises the measurement: ``placement`` (how ``timeit`` placed its arrays:
                      ``"split"`` = outputs from the split allocator, what ``evaluate`` allocates itself;
                      ``"separate"`` = every array from torch).  ``retrieve`` compares facts of one placement only.
``compiler_version``  ``"AMD-ROCm <hip version>"`` instead of the OpenCL vendor/driver.
keys                  the einsum is keyed by the canonical form of
                      ``feinsum_amd.canonicalization`` (exhaustive search), not by the
                      reference's graph-canonical labelling: the two sets of keys do not
                      collide.  Archives WRITTEN BY THE REFERENCE can still be read:
                      :func:`query_reference_archive` re-canonicalises every key of such an
                      archive (the shipped ``data/transform_archive_v2...v5.sqlite`` in
                      their older ``use_matrix`` layout as well as the current layout of
                      ``src/feinsum/sql_utils.py:389-415``) and returns the facts recorded
                      for an einsum -- the reference's own timings, e.g. for a
                      side-by-side table; their ``transform_id`` names a loopy transform
                      file, which this backend cannot run.

:func:`retrieve` closes the loop the reference closes with its archive: it returns the
recorded variant with the highest GFLOP/s as a ``transform`` that ``timeit`` /
``evaluate`` accept.
"""

from __future__ import annotations

import json
import logging
import os
import sqlite3
from dataclasses import dataclass
from types import MappingProxyType
from typing import Any, Callable, Mapping, Optional, Sequence, Tuple, Union

import numpy as np

from feinsum_amd.canonicalization import canonicalize_einsum
from feinsum_amd.contraction_schedule import count_ops
from feinsum_amd.diagnostics import NoFactInDatabaseError
from feinsum_amd.einsum import INT_CLASSES, BatchedEinsum, SizeParam

logger = logging.getLogger(__name__)

#: the archive shipped with the package (facts measured on MI355X; see tools/record_archive.py)
DEFAULT_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data",
                          "transform_archive_mi355x.sqlite")
TIMINGS_TABLENAME = "FEINSUM_TIMING_FACTS"
KNOWN_VARIANTS = ("mfma", "tiled", "generic")


# -- column encodings (reference: sql_utils.py:54-130) ------------------------------------

def dump_arg_to_dtype(einsum: BatchedEinsum) -> str:
    return json.dumps({arg: np.dtype(dtype).name for arg, dtype in einsum.arg_to_dtype.items()},
                      sort_keys=True)


def dump_index_to_length(einsum: BatchedEinsum) -> str:
    return json.dumps({k: int(v) for k, v in einsum.index_to_dim_length.items()
                       if isinstance(v, INT_CLASSES)}, sort_keys=True)


def dump_arg_names(einsum: BatchedEinsum) -> str:
    return json.dumps([[arg.name for arg in row] for row in einsum.args])


def dump_op_info(einsum: BatchedEinsum, long_dim_length: int) -> str:
    # one result dtype per einsum in this build (reference: per-dtype map, sql_utils.py:85-99)
    dtype = np.result_type(*[arg.dtype for arg in einsum.args[0]])
    return json.dumps({dtype.name: count_ops(einsum, long_dim_length=long_dim_length) / 1e9},
                      sort_keys=True)


def load_op_info(op_info: str) -> Mapping[np.dtype, float]:
    return MappingProxyType({np.dtype(k): float(v) for k, v in json.loads(op_info).items()})


def _process_param(param: Any) -> Any:
    if isinstance(param, (int, bool, str)):
        return param
    if isinstance(param, list):
        return tuple(_process_param(k) for k in param)
    raise NotImplementedError(type(param))


def load_transform_params(params_str: str) -> Mapping[str, Any]:
    params = json.loads(params_str)
    assert isinstance(params, dict) and all(isinstance(k, str) for k in params)
    return MappingProxyType({k: _process_param(v) for k, v in params.items()})


def dump_device_name(device: Any) -> str:
    """``device`` is anything with a ``.name`` (``DeviceQueue.device``, ``FakeCLDevice``) or a str."""
    name = device if isinstance(device, str) else device.name
    assert isinstance(name, str)
    return (name.replace(" ", "_").replace("-", "_").replace("@", "AT").replace("(", "_")
            .replace(")", "_").replace(".", "DOT"))


def dump_compiler_version() -> str:
    try:
        import torch

        return f"AMD-ROCm {torch.version.hip}"
    except Exception:   # noqa: BLE001  (torch is optional for reading an archive)
        return "AMD-ROCm unknown"


# -- facts ------------------------------------------------------------------------------

@dataclass(frozen=True)
class QueryInfo:
    transform_id: str
    transform_params: Mapping[str, Any]
    runtime_in_sec: float
    compiler_version: str
    giga_op_info: Mapping[np.dtype, float]
    _einsum: BatchedEinsum

    def giga_op_rate(self, dtype: Any) -> float:
        return self.giga_op_info[np.dtype(dtype)] / self.runtime_in_sec

    @property
    def transform(self) -> Mapping[str, Any]:
        """What to pass as ``transform=`` to ``timeit`` / ``evaluate`` to get this variant."""
        return MappingProxyType({"variant": self.transform_id, **self.transform_params})


def _connect(database: Union[str, sqlite3.Connection]) -> Tuple[sqlite3.Connection, bool]:
    if isinstance(database, sqlite3.Connection):
        return database, False
    return sqlite3.connect(database), True


def _has_table(conn: sqlite3.Connection) -> bool:
    cur = conn.cursor()
    cur.execute("SELECT name FROM sqlite_master WHERE (type='table' AND name=?);", (TIMINGS_TABLENAME,))
    return bool(cur.fetchall())


def _create_timings_table_if_non_existent(conn: sqlite3.Connection) -> None:
    if not _has_table(conn):
        logger.info("Table %s not in DB, creating one.", TIMINGS_TABLENAME)
        conn.cursor().execute(
            f"CREATE TABLE {TIMINGS_TABLENAME} ("
            " ID INTEGER PRIMARY KEY AUTOINCREMENT,"
            " subscripts TEXT, index_to_length TEXT, args TEXT, arg_to_dtype TEXT,"
            " device_name TEXT, transform_id TEXT, transform_params TEXT,"
            " runtime_in_sec REAL, compiler_version TEXT, giga_op_info TEXT, timestamp TEXT)")
    conn.commit()


def _key(einsum: BatchedEinsum) -> Tuple[str, str, str, str]:
    return (einsum.get_subscripts(), dump_index_to_length(einsum), dump_arg_names(einsum),
            dump_arg_to_dtype(einsum))


def query(einsum: BatchedEinsum, device: Any, *, database: Union[str, sqlite3.Connection] = DEFAULT_DB,
          err_if_no_results: bool = False) -> Tuple[QueryInfo, ...]:
    """Facts of previously recorded runs of *einsum* on *device* (reference: sql_utils.py:160-246)."""
    einsum = canonicalize_einsum(einsum)
    conn, own = _connect(database)
    try:
        if not _has_table(conn):
            raise RuntimeError(f"Database '{database}' does not contain the timing facts table.")
        cur = conn.cursor()
        cur.execute(
            "SELECT transform_id, transform_params, runtime_in_sec, compiler_version, giga_op_info"
            f" FROM {TIMINGS_TABLENAME} WHERE (subscripts = ? AND index_to_length = ? AND args = ?"
            " AND arg_to_dtype = ? AND device_name = ?);",
            (*_key(einsum), dump_device_name(device)))
        facts = cur.fetchall()
    finally:
        if own:
            conn.close()
    result = tuple(QueryInfo(f[0], load_transform_params(f[1]), f[2], f[3], load_op_info(f[4]), einsum)
                   for f in facts)
    if not result and err_if_no_results:
        sizes = ", ".join(f"{idx}: {n}" for idx, n in einsum.index_to_dim_length.items()
                          if not isinstance(n, SizeParam))
        raise NoFactInDatabaseError(
            f"No facts found for the einsum: `{einsum.get_subscripts()} [{sizes}] [#outputs={einsum.b}]`.")
    return result


def retrieve(einsum: BatchedEinsum, device: Any, *, database: Union[str, sqlite3.Connection] = DEFAULT_DB,
             consider_query: Optional[Callable[[QueryInfo], bool]] = None) -> Mapping[str, Any]:
    """The recorded variant with the highest op rate, as a ``transform`` (reference: sql_utils.py:249-294)."""
    consider = consider_query or (lambda q: True)
    queries = [q for q in query(einsum, device, database=database, err_if_no_results=True) if consider(q)]
    if not queries:
        raise NoFactInDatabaseError(f"No facts found for the einsum: `{einsum}`, with the filtering"
                                    f" function: {consider_query!r}.")
    # Facts taken with different placements of the timed arrays are not comparable (the same kernel is 5-12 % apart between
    # allocator arrays and torch arrays): variants are ranked within ONE placement -- the one evaluate() uses for the
    # outputs it allocates ("split") when any fact has it, else the placement with the most facts.
    by_placement: dict = {}
    for q in queries:
        by_placement.setdefault(q.transform_params.get("placement", "separate"), []).append(q)
    pool = by_placement.get("split") or max(by_placement.values(), key=len)
    best = max(pool, key=lambda q: sum(q.giga_op_rate(dt) for dt in q.giga_op_info))
    return best.transform


def _get_batched_einsum_from_sql_row(subscripts: str, index_to_length: Mapping[str, Any],
                                     arg_names: Sequence[Sequence[str]],
                                     arg_to_dtype: Mapping[str, str]) -> BatchedEinsum:
    from feinsum_amd.make_einsum import array, batched_einsum

    in_specs = subscripts.split("->")[0]
    in_idx_sets = [tuple(spec.strip()) for spec in in_specs.split(",")]
    lengths = dict(index_to_length)
    for idx_set in in_idx_sets:
        for idx in idx_set:
            lengths.setdefault(idx, SizeParam(idx.upper()))
    shapes = {arg: tuple(lengths[i] for i in idx_set)
              for row in arg_names for idx_set, arg in zip(in_idx_sets, row)}
    return batched_einsum(subscripts, [[array(a, shapes[a], arg_to_dtype[a]) for a in row]
                                       for row in arg_names])


def get_timed_einsums_in_db(device: Any, database: Union[str, sqlite3.Connection] = DEFAULT_DB
                            ) -> Tuple[BatchedEinsum, ...]:
    """Einsums with at least one fact for *device* (reference: sql_utils.py:342-386)."""
    conn, own = _connect(database)
    try:
        cur = conn.cursor()
        cur.execute(f"SELECT subscripts, index_to_length, args, arg_to_dtype FROM {TIMINGS_TABLENAME}"
                    " WHERE device_name = ?;", (dump_device_name(device),))
        facts = sorted(set(cur.fetchall()))
    finally:
        if own:
            conn.close()
    seen = [_get_batched_einsum_from_sql_row(s, json.loads(l), json.loads(a), json.loads(d))
            for s, l, a, d in facts]
    assert len(set(seen)) == len(seen)   # the canonical form is sound
    return tuple(seen)


def _timestamp() -> str:
    from datetime import datetime

    try:
        import pytz

        now = datetime.now(pytz.timezone("America/Chicago"))   # as the reference: sql_utils.py:463-469
    except ImportError:
        now = datetime.now()
    return now.strftime("%Y_%m_%d_%H%M%S")


def record_facts(einsum: BatchedEinsum, cq: Any, variant: str = "mfma",
                 transform_params: Optional[Mapping[str, Any]] = None,
                 database: Union[str, sqlite3.Connection] = DEFAULT_DB,
                 long_dim_length: int = 100_000, *, runtime_in_sec: Optional[float] = None,
                 device_name: Optional[str] = None) -> None:
    """
    Time *einsum* with kernel variant *variant* on *cq* (``timeit``, the reference's
    protocol) and store the fact (reference: sql_utils.py:413-509).  ``runtime_in_sec``
    / ``device_name`` record a measurement taken elsewhere instead of timing here.
    """
    params = dict(transform_params or {})
    if variant not in KNOWN_VARIANTS:
        raise ValueError(f"unknown kernel variant '{variant}' (known: {KNOWN_VARIANTS})")
    einsum = canonicalize_einsum(einsum)
    if runtime_in_sec is None:
        from feinsum_amd.measure import _as_queue, timeit_details

        timing = timeit_details(einsum, cq=cq, transform={"variant": variant, **params},
                                long_dim_length=long_dim_length)
        runtime_in_sec = timing.seconds_device
        # how the timed arrays were placed is part of the fact (transform_params is the reference's JSON column for
        # what parametrises a measurement): "split" = outputs from the split allocator (what evaluate() allocates itself),
        # "separate" = every array from torch (what a caller who hands in torch.empty outputs gets)
        params["placement"] = timing.placement.get("mode", "separate")
        if timing.placement.get("fallback"):
            params["placement_fallback"] = timing.placement["fallback"]
        device_name = device_name or _as_queue(cq).device.name
    if device_name is None:
        raise ValueError("device_name is needed with an external runtime_in_sec")
    conn, own = _connect(database)
    try:
        _create_timings_table_if_non_existent(conn)
        conn.cursor().execute(
            f"INSERT INTO {TIMINGS_TABLENAME} (subscripts, index_to_length, args, arg_to_dtype,"
            " device_name, transform_id, transform_params, runtime_in_sec, compiler_version,"
            " giga_op_info, timestamp) VALUES (?,?,?,?,?,?,?,?,?,?,?)",
            (*_key(einsum), dump_device_name(device_name), variant, json.dumps(params, sort_keys=True),
             float(runtime_in_sec), dump_compiler_version(), dump_op_info(einsum, long_dim_length),
             _timestamp()))
        conn.commit()
    finally:
        if own:
            conn.close()


# -- reading archives written by the reference ------------------------------------------

def _einsum_of_reference_key(subscripts: str, index_to_length: Mapping[str, int], names: Sequence[Sequence[Any]],
                             dtypes: Mapping[str, str]) -> BatchedEinsum:
    """A reference archive key as a BatchedEinsum.  *names*: the argument-name matrix, one row per batch member; an
    entry is a name (current layout, ``args``) or a list of value names (older ``use_matrix`` layout: an array access
    that multiplies several values becomes one operand per value with the same indices)."""
    from feinsum_amd.make_einsum import array, batched_einsum

    ins, out = subscripts.replace(" ", "").split("->")
    in_sets = ins.split(",")
    shape = lambda idxs: tuple(index_to_length.get(ch, ch.upper()) for ch in idxs)   # noqa: E731
    as_list = lambda entry: list(entry) if isinstance(entry, (list, tuple)) else [entry]   # noqa: E731
    subs = ",".join(s for s, uses in zip(in_sets, names[0]) for _ in as_list(uses)) + "->" + out
    args = [[array(name, shape(s), dtypes[name]) for s, uses in zip(in_sets, row) for name in as_list(uses)] for row in names]
    return batched_einsum(subs, args)


def query_reference_archive(einsum: BatchedEinsum, database: Union[str, sqlite3.Connection], *,
                            device_name: Optional[str] = None) -> Tuple[QueryInfo, ...]:
    """
    The facts an archive WRITTEN BY THE REFERENCE holds for *einsum* (any naming, any operand order): every distinct
    key of the archive is rebuilt as a :class:`BatchedEinsum` and compared in this build's canonical form.  Reads the
    shipped ``data/transform_archive_v2...v5.sqlite`` (one table per device in v2, columns ``use_matrix`` /
    ``value_to_dtype``) and the current layout (``FEINSUM_TIMING_FACTS`` with ``args`` / ``arg_to_dtype``,
    ``src/feinsum/sql_utils.py:389-415``).  *device_name*: keep only facts of that device (mangled as the reference
    does, e.g. ``"NVIDIA_TITAN_V"``; for v2 the table name).  The returned ``QueryInfo.transform_id`` is the reference's
    transform file; ``giga_op_rate`` works as for facts of this build.
    """
    want = _key(canonicalize_einsum(einsum))
    conn, own = _connect(database)
    hits = []
    try:
        tables = [name for (name,) in conn.execute("select name from sqlite_master where type='table'")
                  if not name.startswith("sqlite_")]
        for table in tables:
            cols = [row[1] for row in conn.execute(f"pragma table_info({table})")]
            if "subscripts" not in cols or "runtime_in_sec" not in cols:
                continue
            names_col = "args" if "args" in cols else "use_matrix"
            dtype_col = "arg_to_dtype" if "arg_to_dtype" in cols else "value_to_dtype"
            device_expr = "device_name" if "device_name" in cols else f"'{table}'"
            params_expr = "transform_params" if "transform_params" in cols else "'{}'"
            compiler_expr = "compiler_version" if "compiler_version" in cols else "''"
            matching = {}
            for subs, i2l, names, dts in conn.execute(
                    f"select distinct subscripts, index_to_length, {names_col}, {dtype_col} from {table}"):
                try:
                    key = _key(canonicalize_einsum(_einsum_of_reference_key(subs, json.loads(i2l), json.loads(names),
                                                                            json.loads(dts))))
                except (ValueError, TypeError, NotImplementedError, AssertionError):
                    continue        # a key this build's builders do not accept: not a match for anything it can evaluate
                if key == want:
                    matching[(subs, i2l, names, dts)] = True
            for subs, i2l, names, dts in matching:
                for tid, params, runtime, compiler, gops, dev in conn.execute(
                        f"select transform_id, {params_expr}, runtime_in_sec, {compiler_expr}, giga_op_info, {device_expr} "
                        f"from {table} where subscripts = ? and index_to_length = ? and {names_col} = ? and {dtype_col} = ?",
                        (subs, i2l, names, dts)):
                    if device_name is not None and dev != device_name:
                        continue
                    hits.append(QueryInfo(tid, load_transform_params(params), float(runtime), compiler or "",
                                          load_op_info(gops), einsum))
    finally:
        if own:
            conn.close()
    return tuple(hits)
