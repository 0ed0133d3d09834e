"""
Contraction schedules and the algorithmic operation counter.

API mirror of the reference's ``feinsum.contraction_schedule`` (reference:
``src/feinsum/contraction_schedule.py:34-58`` Argument types, ``:61-98``
ContractionSchedule, ``:101-110`` trivial schedule, ``:113-178`` opt_einsum
schedule).  The reference delegates the search to ``opt_einsum.contract_path
(optimize="optimal")``; opt_einsum is not a dependency here, so
:func:`get_opt_einsum_contraction_schedule` runs its own exhaustive search over
pairwise contraction orders minimising the same cost (operation count with
every ``SizeParam`` set to ``long_dim_length``, default 1e6).  For the DG
family it returns the schedules the survey records:

* grad   ``ej,rij->rie``  then ``rie,xre->xei``
* div    ``xej,xre->rje`` then ``rje,rij->ei``
* lift   ``fej,ef->fje``  then ``fje,fij->ei``

:func:`count_ops` is the build's stand-in for the reference's op counter
(``measure.py:278-331``: loopy ``get_op_map`` on the optimal schedule, adds +
muls, an N-term reduction counted as N adds): per step and per output entry,
``(n_operands - 1) * N`` multiplies plus ``N`` adds when the step reduces over
N > 0 summation points... pinned by the reference's known answers -- grad p4
trivial 33075 / optimal 7980 per element (``test/test_loopy_utils.py:270-271``)
and face-mass x4 17040 (archive ``giga_op_info`` 1.704 at E = 1e5).
"""

from __future__ import annotations

from dataclasses import dataclass, replace
from itertools import combinations
from typing import Any, Dict, List, Mapping, Sequence, Tuple, Union

from feinsum_amd.einsum import BatchedEinsum, SizeParam


@dataclass(frozen=True)
class EinsumOperand:
    """The *ioperand*-th operand of the original einsum."""

    ioperand: int


@dataclass(frozen=True)
class IntermediateResult:
    """The result of an earlier step, by name."""

    name: str


Argument = Union[EinsumOperand, IntermediateResult]


@dataclass(frozen=True, eq=True, repr=True)
class ContractionSchedule:
    """
    A series of einsums: step ``i`` evaluates ``subscripts[i]`` on
    ``arguments[i]`` and names its result ``result_names[i]``.
    """

    subscripts: Tuple[str, ...]
    result_names: Tuple[str, ...]
    arguments: Tuple[Tuple[Argument, ...], ...]

    def __post_init__(self) -> None:
        assert len(self.subscripts) == len(self.result_names) == len(self.arguments)

    @property
    def nsteps(self) -> int:
        return len(self.subscripts)

    def copy(self, **kwargs: Any) -> "ContractionSchedule":
        return replace(self, **kwargs)


def get_trivial_contraction_schedule(einsum: BatchedEinsum) -> ContractionSchedule:
    """The whole einsum as one contraction."""
    return ContractionSchedule(
        (einsum.get_subscripts(),),
        ("_fe_out",),
        (tuple(EinsumOperand(i) for i in range(einsum.n)),),
    )


def _dim_lengths(expr: BatchedEinsum, long_dim_length: int) -> Dict[str, int]:
    return {idx: (long_dim_length if isinstance(d, SizeParam) else int(d))
            for idx, d in expr.index_to_dim_length.items()}


def _step_ops(operand_idx_sets: Sequence[frozenset], out_idxs: frozenset,
              dims: Mapping[str, int]) -> int:
    """mul + add count of one contraction step (see module docstring)."""
    all_idxs = frozenset().union(*operand_idx_sets)
    n_points = 1
    for idx in all_idxs:
        n_points *= dims[idx]
    n_mults = (len(operand_idx_sets) - 1) * n_points
    n_adds = n_points if (all_idxs - out_idxs) else 0
    return n_mults + n_adds


def _optimal_pairwise_order(idx_sets: List[frozenset], out_idxs: frozenset,
                            dims: Mapping[str, int]):
    """Exhaustive DFS over pairwise contraction orders (n <= ~7 operands)."""
    best: Dict[str, Any] = {"cost": None, "path": None}

    def needed_after(remaining: List[frozenset]) -> frozenset:
        s = set(out_idxs)
        for r in remaining:
            s |= r
        return frozenset(s)

    def rec(terms: List[Tuple[frozenset, Any]], cost: int, path: list) -> None:
        if best["cost"] is not None and cost >= best["cost"]:
            return
        if len(terms) == 1:
            best["cost"], best["path"] = cost, list(path)
            return
        for a, b in combinations(range(len(terms)), 2):
            rest = [t for k, t in enumerate(terms) if k not in (a, b)]
            keep = (terms[a][0] | terms[b][0]) & needed_after([t[0] for t in rest])
            if len(terms) == 2:
                keep = out_idxs
            step_cost = _step_ops([terms[a][0], terms[b][0]], keep, dims)
            path.append((terms[a][1], terms[b][1], keep))
            rec(rest + [(keep, ("tmp", len(path) - 1))], cost + step_cost, path)
            path.pop()

    rec([(s, ("op", i)) for i, s in enumerate(idx_sets)], 0, [])
    return best["cost"], best["path"]


def get_opt_einsum_contraction_schedule(expr: BatchedEinsum,
                                        **opt_einsum_kwargs: Any) -> ContractionSchedule:
    """
    The operation-count-optimal pairwise schedule (what
    ``opt_einsum.contract_path(optimize="optimal")`` returns for the reference,
    ``contraction_schedule.py:113-178``).  Accepted kwargs: ``long_dim_length``
    (default 1_000_000); ``optimize`` / ``use_blas`` are accepted and ignored.
    """
    long_dim_length = int(opt_einsum_kwargs.pop("long_dim_length", 1_000_000))
    opt_einsum_kwargs.pop("optimize", None)
    opt_einsum_kwargs.pop("use_blas", None)
    if opt_einsum_kwargs:
        raise TypeError(f"unexpected arguments: {sorted(opt_einsum_kwargs)}")

    dims = _dim_lengths(expr, long_dim_length)
    idx_sets = [frozenset(s) for s in expr.in_idx_sets]
    out = frozenset(expr.out_idx_set)
    if expr.n == 1:
        return get_trivial_contraction_schedule(expr)
    # a single n-ary contraction is a candidate too (it wins e.g. for pure
    # outer products); keep it when no pairwise order beats it
    trivial_cost = _step_ops(idx_sets, out, dims)
    cost, path = _optimal_pairwise_order(idx_sets, out, dims)
    if cost is None or trivial_cost <= cost:
        return get_trivial_contraction_schedule(expr)

    in_strs = ["".join(s) for s in expr.in_idx_sets]
    tmp_strs: Dict[int, str] = {}
    subscripts, result_names, arguments = [], [], []
    for istep, (a, b, keep) in enumerate(path):
        def _describe(term):
            kind, k = term
            if kind == "op":
                return in_strs[k], EinsumOperand(k)
            return tmp_strs[k], IntermediateResult(result_names[k])

        sa, arga = _describe(a)
        sb, argb = _describe(b)
        last = istep == len(path) - 1
        if last:
            out_str = "".join(expr.out_idx_set)
        else:  # deterministic index order: first appearance in the two operands
            out_str = "".join(dict.fromkeys(ch for ch in sa + sb if ch in keep))
        tmp_strs[istep] = out_str
        subscripts.append(f"{sa},{sb}->{out_str}")
        result_names.append("_fe_out" if last else ("_fe_tmp" if istep == 0 else f"_fe_tmp_{istep - 1}"))
        arguments.append((arga, argb))
    return ContractionSchedule(tuple(subscripts), tuple(result_names), tuple(arguments))


def count_ops(expr: BatchedEinsum, schedule: ContractionSchedule | None = None,
              long_dim_length: int = 1) -> int:
    """
    Operation count (adds + muls, all ``b`` einsums) of evaluating *expr* with
    *schedule* (default: the optimal one), with every ``SizeParam`` set to
    *long_dim_length*.  It is linear in that length whenever the long index
    survives to the output, so ``count_ops(expr, long_dim_length=1)`` is the
    per-element count: 7980 for grad/div p4, 17040 for face-mass x4.
    """
    if schedule is None:
        schedule = get_opt_einsum_contraction_schedule(expr)
    dims = _dim_lengths(expr, long_dim_length)
    total = 0
    for subs in schedule.subscripts:
        lhs, rhs = subs.replace(" ", "").split("->")
        total += _step_ops([frozenset(s) for s in lhs.split(",")], frozenset(rhs), dims)
    return total * expr.b
