"""
Canonical form of a :class:`BatchedEinsum`: one representative per class of einsums
that differ only in the names of indices, arrays and size parameters and in the order
of the operands.

The reference gets its canonical form from a graph-canonical labelling (pybliss;
``src/feinsum/canonicalization.py:1087`` ``canonicalize_einsum``) and uses it as the key
of the transform archive (``src/feinsum/sql_utils.py:176,407``).  The einsums this build
evaluates have at most a handful of operands, so the representative is found by
exhaustive search instead: for every operand order the indices are renamed ``a, b, c, ...``
in order of first use (output first, then the operands left to right), the arrays
``arg_0, arg_1, ...`` in order of first use over the rows, and the lexicographically
smallest description wins.  The labels therefore differ from the reference's: keys
written by this build and keys written by the reference do not collide, and are not
meant to be looked up from one another (see ``feinsum_amd.sql_utils``).

Row (output) order is part of an einsum's meaning for its caller and is kept.
"""

from __future__ import annotations

from itertools import permutations
from typing import Dict, List, Tuple

from feinsum_amd.einsum import Array, BatchedEinsum, SizeParam

_LETTERS = "abcdefghijklmnopqrstuvwxyz"
MAX_OPERANDS_FOR_SEARCH = 8


def _describe(einsum: BatchedEinsum, perm: Tuple[int, ...]):
    index_name: Dict[str, str] = {}
    for idx in list(einsum.out_idx_set) + [i for p in perm for i in einsum.in_idx_sets[p]]:
        if idx not in index_name:
            index_name[idx] = _LETTERS[len(index_name)]
    size_name: Dict[str, str] = {}
    dims: List[Tuple[str, object]] = []
    for idx, new in sorted(index_name.items(), key=lambda kv: kv[1]):
        d = einsum.index_to_dim_length[idx]
        if isinstance(d, SizeParam):
            dims.append((new, size_name.setdefault(d.name, new.upper())))
        else:
            dims.append((new, int(d)))
    arg_name: Dict[str, str] = {}
    rows = []
    for row in einsum.args:
        names = []
        for p in perm:
            names.append(arg_name.setdefault(row[p].name, f"arg_{len(arg_name)}"))
        rows.append(tuple(names))
    subscripts = (",".join("".join(index_name[i] for i in einsum.in_idx_sets[p]) for p in perm)
                  + "->" + "".join(index_name[i] for i in einsum.out_idx_set))
    dtypes = tuple(sorted((new, str(einsum.arg_to_dtype[old])) for old, new in arg_name.items()))
    key = (subscripts, tuple((n, str(v)) for n, v in dims), tuple(rows), dtypes)
    return key, index_name, size_name, arg_name


def canonicalize_einsum(einsum: BatchedEinsum) -> BatchedEinsum:
    """The canonical representative of *einsum* (idempotent)."""
    if einsum.n > MAX_OPERANDS_FOR_SEARCH:
        raise NotImplementedError(f"canonical form by exhaustive search is limited to"
                                  f" {MAX_OPERANDS_FOR_SEARCH} operands (got {einsum.n})")
    best = None
    for perm in permutations(range(einsum.n)):
        cand = _describe(einsum, perm)
        if best is None or cand[0] < best[0][0]:
            best = (cand, perm)
    (_, index_name, size_name, arg_name), perm = best

    def shape_of(arr: Array):
        return tuple(SizeParam(size_name[d.name]) if isinstance(d, SizeParam) else d for d in arr.shape)

    args = tuple(tuple(Array(arg_name[row[p].name], shape_of(row[p]), row[p].dtype) for p in perm)
                 for row in einsum.args)
    return BatchedEinsum(
        out_idx_set=tuple(index_name[i] for i in einsum.out_idx_set),
        in_idx_sets=tuple(tuple(index_name[i] for i in einsum.in_idx_sets[p]) for p in perm),
        args=args)
