"""
CPU oracle for the DG-wave batched-einsum hot path.  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg --
never by the product package (feinsum_amd/ has no CPU fallback).

What it restates
----------------
The reference defines "correct" for a transformed kernel as agreement with

    np.einsum(einsum.get_subscripts(), *inputs_of_row, optimize="optimal")

for every row of the batched einsum, atol = rtol = 1e-10 in float64
(reference: src/feinsum/measure.py:149-159 for the evaluation, :178-192 for
the tolerances).  numpy is available wherever this build runs, so
:func:`reference_outputs` is that very expression -- not an approximation of it.
The semantics being evaluated are those of the loop nest the reference emits,
``out[free...] = sum_{summed...} prod_k arg_k[idx_k]`` with the result dtype
``np.result_type`` of the operands (reference: src/feinsum/codegen/loopy.py:242-305,
:258-260).

Pinning
--------------------
The reference cannot be imported here: it needs Python >= 3.12 syntax
(pyproject.toml:15; loopy_utils/__init__.py:567) and loopy / pyopencl / islpy /
pymbolic / opt_einsum / immutables, none of which are installed (ordinary
ImportError / SyntaxError, not a permission denial), and it ships no stored
golden vectors: every value check in its test-suite recomputes np.einsum on the
fly (test/test_codegen.py:34-120 via measure.py:111-194).  The oracle is
therefore pinned by
  (1) being the reference's own ground-truth expression (above);
  (2) :func:`naive_longdouble`, an independent sum-of-products evaluation in
      extended precision that does not go through np.einsum;
  (3) oracle/loopnest.c, a plain C restatement of the same loop nests;
  (4) the reference's known-answer integers for this path, checked in
      tests/test_schedule_opcount.py: 33075 / 7980 flops per element for grad p4
      (test/test_loopy_utils.py:270-271), the 2-step optimal schedule
      (test/test_codegen.py:134-137), 0.798 / 0.798 / 1.704 GFLOP at E = 1e5
      from data/transform_archive_v5.sqlite.
Golden vectors produced from (1), cross-checked with (2), are committed under
tests/golden/ together with the script that made them.
"""

from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np


def _split(subscripts: str):
    lhs, rhs = subscripts.replace(" ", "").split("->")
    return [tuple(s) for s in lhs.split(",")], tuple(rhs)


def reference_outputs(subscripts: str, rows: Sequence[Sequence[np.ndarray]]) -> List[np.ndarray]:
    """The reference's ground truth, one output per row (measure.py:149-159)."""
    return [np.einsum(subscripts, *row, optimize="optimal") for row in rows]


def naive_longdouble(subscripts: str, operands: Sequence[np.ndarray]) -> np.ndarray:
    """
    Independent evaluation: broadcast every operand onto the full index space in
    ``np.longdouble``, multiply, and sum the contracted axes with pairwise
    summation -- the literal sum-of-products of codegen/loopy.py:289-305, without
    np.einsum.  Memory = product of ALL index extents, so small sizes only.
    """
    in_sets, out_set = _split(subscripts)
    all_idx = sorted(set(i for s in in_sets for i in s))
    pos = {idx: k for k, idx in enumerate(all_idx)}
    prod = None
    for arr, idxs in zip(operands, in_sets):
        a = np.asarray(arr, dtype=np.longdouble)
        # move operand axes into sorted-index order, then insert broadcast axes
        order = sorted(range(len(idxs)), key=lambda ax: pos[idxs[ax]])
        a = a.transpose(order)
        shape = [1] * len(all_idx)
        for ax in order:
            shape[pos[idxs[ax]]] = arr.shape[ax]
        a = a.reshape(shape)
        prod = a if prod is None else prod * a
    sum_axes = tuple(pos[i] for i in all_idx if i not in out_set)
    res = prod.sum(axis=sum_axes) if sum_axes else prod
    kept = [i for i in all_idx if i in out_set]
    res = res.transpose([kept.index(i) for i in out_set])
    return res


def loop_reference(subscripts: str, operands: Sequence[np.ndarray]) -> np.ndarray:
    """Pure-Python nested loops in float64 (tiny cases only): the loop nest verbatim."""
    from itertools import product

    in_sets, out_set = _split(subscripts)
    extent: Dict[str, int] = {}
    for arr, idxs in zip(operands, in_sets):
        for ax, idx in enumerate(idxs):
            extent[idx] = arr.shape[ax]
    sum_idx = [i for i in dict.fromkeys(i for s in in_sets for i in s) if i not in out_set]
    out = np.zeros([extent[i] for i in out_set],
                   dtype=np.result_type(*[np.asarray(o).dtype for o in operands]))
    for o in product(*[range(extent[i]) for i in out_set]):
        env = dict(zip(out_set, o))
        acc = 0.0
        for s in product(*[range(extent[i]) for i in sum_idx]):
            env.update(zip(sum_idx, s))
            term = 1.0
            for arr, idxs in zip(operands, in_sets):
                term = term * arr[tuple(env[i] for i in idxs)]
            acc += term
        out[o] = acc
    return out


def max_rel_err(got: np.ndarray, ref: np.ndarray) -> float:
    """max |got - ref| / max |ref|  (the normwise measure of SURVEY H4)."""
    ref = np.asarray(ref)
    denom = float(np.max(np.abs(ref))) if ref.size else 1.0
    if ref.size == 0:
        return 0.0
    return float(np.max(np.abs(np.asarray(got, dtype=np.longdouble) - ref))) / (denom or 1.0)
