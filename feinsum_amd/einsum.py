"""
Immutable description of a batched einsum -- the IR every other module consumes.

API mirror of the reference's ``feinsum.einsum`` (reference:
``src/feinsum/einsum.py:26-41`` SizeParam, ``:48-83`` Array, ``:86-124`` axis
access tags, ``:127-387`` BatchedEinsum).  Differences, all deliberate:

* mappings are read-only views of plain dicts (``types.MappingProxyType``)
  whose iteration order is the deterministic first-use order, instead of
  ``immutables.Map`` (hash order, process dependent -- SURVEY H5);
* no third-party dependencies (numpy only; ``tabulate`` only for ``__str__``);
* ``__str__`` prints the iteration domain in ISL-like notation without islpy.
"""

from __future__ import annotations

from dataclasses import dataclass, replace
from functools import cached_property
from types import MappingProxyType
from typing import Any, Mapping, Tuple, Union

import numpy as np

IntegralT = Union[int, np.integer]
INT_CLASSES = (int, np.integer)


@dataclass(frozen=True)
class SizeParam:
    """A parametric ("long") axis length, e.g. the number of elements ``E``."""

    name: str

    def __truediv__(self, other: Any) -> Any:
        # the reference keeps this only so tuner parameter getters can be
        # written uniformly (einsum.py:36-41); arithmetic is not supported.
        return NotImplemented

    __rtruediv__ = __truediv__


ShapeComponentT = Union[int, np.integer, SizeParam]
ShapeT = Tuple[ShapeComponentT, ...]


@dataclass(frozen=True, eq=True, repr=True)
class Array:
    """A named multidimensional array operand (name, shape, dtype)."""

    name: str
    shape: ShapeT
    dtype: np.dtype

    @property
    def ndim(self) -> int:
        return len(self.shape)

    def copy(self, *, name: str | None = None, shape: ShapeT | None = None,
             dtype: np.dtype | None = None) -> "Array":
        return replace(
            self,
            name=self.name if name is None else name,
            shape=self.shape if shape is None else shape,
            dtype=self.dtype if dtype is None else dtype,
        )


@dataclass(frozen=True)
class EinsumAxisAccess:
    """Base class of the per-index access tags; abstract."""

    def __init__(self) -> None:
        if type(self) is EinsumAxisAccess:
            raise TypeError("EinsumAxisAccess is abstract and cannot be instantiated directly")


@dataclass(frozen=True)
class FreeAxis(EinsumAxisAccess):
    """Index that survives into the output at position *output_index*."""

    output_index: int


@dataclass(frozen=True)
class SummationAxis(EinsumAxisAccess):
    """Index that is contracted; *index* numbers the reduction indices."""

    index: int


def _frozen(d: dict) -> Mapping:
    return MappingProxyType(dict(d))


@dataclass(frozen=True)
class BatchedEinsum:
    """
    ``b`` einsums that share one subscript expression (and possibly operands):
    ``out_idx_set`` / ``in_idx_sets`` are tuples of single lower-case letters,
    ``args[b][n]`` the operand matrix.  Construction validates exactly what the
    reference validates (einsum.py:159-196) and raises ``AssertionError`` with
    the same messages; :func:`feinsum_amd.make_einsum.batched_einsum` converts
    those to ``TypeError``.
    """

    out_idx_set: Tuple[str, ...]
    in_idx_sets: Tuple[Tuple[str, ...], ...]
    args: Tuple[Tuple[Array, ...], ...]

    def __post_init__(self) -> None:
        def _ok(idx: Any) -> bool:
            return isinstance(idx, str) and len(idx) == 1 and idx.islower()

        assert all(_ok(i) for i in self.out_idx_set), \
            "Obtained invalid output index (RHS of ->)."
        assert all(_ok(i) for s in self.in_idx_sets for i in s), \
            "Obtained invalid input index (LHS of ->)."
        in_indices = frozenset(i for s in self.in_idx_sets for i in s)
        assert frozenset(self.out_idx_set) <= in_indices, \
            "Obtained an out index which is not present in the input indices."
        assert all(len(row) == len(self.in_idx_sets) for row in self.args), \
            "Mismatch in #operands between subscript expression and input arrays."
        assert all(arg.ndim == len(idxs)
                   for row in self.args for arg, idxs in zip(row, self.in_idx_sets)), \
            "Dimensionality of input operands do no match the provided subscripts."

        # force the consistency checks hidden in the derived maps
        _ = self.arg_to_dtype
        _ = self.arg_to_shape
        _ = self.index_to_dim_length
        names = (set(self.all_args) | set(self.all_indices)
                 | {p.name for p in self.all_size_params})
        assert (len(self.all_args) + len(self.all_indices)
                + len(self.all_size_params)) == len(names), \
            "Must use different names for arguments, indices, and size params."

    # -- sizes -------------------------------------------------------------
    @cached_property
    def b(self) -> int:
        """Number of einsums in the batch."""
        return len(self.args)

    @cached_property
    def n(self) -> int:
        """Number of operands of each einsum."""
        return len(self.in_idx_sets)

    @cached_property
    def index_to_dim_length(self) -> Mapping[str, ShapeComponentT]:
        result: dict = {}
        for row in self.args:
            for arg, idxs in zip(row, self.in_idx_sets):
                for length, idx in zip(arg.shape, idxs):
                    if result.setdefault(idx, length) != length:
                        raise AssertionError("Shape mismatch for indices across the arguments.")
        return _frozen(result)

    @cached_property
    def shape(self) -> ShapeT:
        """Shape of each output."""
        return tuple(self.index_to_dim_length[i] for i in self.out_idx_set)

    @property
    def ndim(self) -> int:
        return len(self.shape)

    def get_subscripts(self) -> str:
        """The subscript expression, e.g. ``'xre,rij,ej -> xei'``."""
        lhs = ",".join("".join(s) for s in self.in_idx_sets)
        return f"{lhs} -> {''.join(self.out_idx_set)}"

    # -- operands ----------------------------------------------------------
    @cached_property
    def arg_to_shape(self) -> Mapping[str, ShapeT]:
        result: dict = {}
        for row in self.args:
            for arg in row:
                if result.setdefault(arg.name, arg.shape) != arg.shape:
                    raise AssertionError(f"Inconsistent shapes for arg {arg.name}.")
        return _frozen(result)

    @cached_property
    def arg_to_dtype(self) -> Mapping[str, np.dtype]:
        result: dict = {}
        for row in self.args:
            for arg in row:
                if result.setdefault(arg.name, arg.dtype) != arg.dtype:
                    raise AssertionError(f"Inconsistent dtypes for arg {arg.name}.")
        return _frozen(result)

    @cached_property
    def index_to_access_descr(self) -> Mapping[str, EinsumAxisAccess]:
        result: dict = {idx: FreeAxis(pos) for pos, idx in enumerate(self.out_idx_set)}
        n_redn = 0
        for idxs in self.in_idx_sets:
            for idx in idxs:
                if idx not in result:
                    result[idx] = SummationAxis(n_redn)
                    n_redn += 1
        return _frozen(result)

    @cached_property
    def sum_indices(self) -> Tuple[str, ...]:
        """Contraction indices in order of first appearance."""
        pairs = [(acc.index, idx) for idx, acc in self.index_to_access_descr.items()
                 if isinstance(acc, SummationAxis)]
        return tuple(idx for _, idx in sorted(pairs))

    @cached_property
    def all_args(self) -> frozenset:
        return frozenset(self.arg_to_shape)

    @cached_property
    def all_indices(self) -> frozenset:
        return frozenset(self.index_to_dim_length)

    @cached_property
    def all_size_params(self) -> frozenset:
        return frozenset(v for v in self.index_to_dim_length.values()
                         if isinstance(v, SizeParam))

    @property
    def output_names(self) -> Tuple[str, ...]:
        """``_fe_out, _fe_out_0, ...`` (reference: measure.py:147, codegen/loopy.py:257)."""
        return ("_fe_out",) + tuple(f"_fe_out_{i}" for i in range(self.b - 1))

    def copy(self, *, out_idx_set=None, in_idx_sets=None, args=None) -> "BatchedEinsum":
        return replace(
            self,
            out_idx_set=self.out_idx_set if out_idx_set is None else out_idx_set,
            in_idx_sets=self.in_idx_sets if in_idx_sets is None else in_idx_sets,
            args=self.args if args is None else args,
        )

    def _domain_str(self) -> str:
        params = sorted(p.name for p in self.all_size_params)
        idxs = sorted(self.all_indices)
        bounds = " and ".join(
            f"0 <= {i} < {d.name if isinstance(d, SizeParam) else int(d)}"
            for i, d in ((i, self.index_to_dim_length[i]) for i in idxs))
        return f"[{', '.join(params)}] -> {{ [{', '.join(idxs)}] : {bounds} }}"

    def __str__(self) -> str:
        from tabulate import tabulate

        rule = "-" * 75
        dtypes = "\n".join(f"{name}: {dt}" for name, dt in sorted(self.arg_to_dtype.items()))
        sum_idxs = "{" + ", ".join(self.sum_indices) + "}"
        out_idxs = ", ".join(self.out_idx_set)
        rows = []
        for out_name, row in zip(self.output_names, self.args):
            product = "×".join(
                f"{arg.name}[{', '.join(idxs)}]" for idxs, arg in zip(self.in_idx_sets, row))
            rows.append([" ", f"{out_name}[{out_idxs}]", "<-", f"Σ_{sum_idxs} {product}"])
        statements = tabulate(rows, tablefmt="plain",
                              colalign=("left", "right", "left", "left"))
        return "\n".join([
            rule, "DOMAINS:", self._domain_str(), rule, "Data-types:", dtypes, rule,
            f"for {','.join(self.out_idx_set)}", statements, "end", rule])
