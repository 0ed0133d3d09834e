"""
numpy-einsum-like builders for :class:`~feinsum_amd.einsum.BatchedEinsum`.

API mirror of the reference's ``feinsum.make_einsum`` (reference:
``src/feinsum/make_einsum.py:55-77`` shape normalisation / ``array``,
``:80-111`` subscript parser, ``:114-148`` ``batched_einsum``, ``:151-156``
``einsum``): explicit-mode subscripts only, a ``str`` axis length becomes a
:class:`SizeParam`, and the error behaviour is the reference's --
``ValueError`` for a missing ``->``, an unparsable character or a repeated
output index, ``NotImplementedError`` for ``...``, ``TypeError`` for any
operand/subscript inconsistency.
"""

from __future__ import annotations

from collections.abc import Iterable, Sequence
from typing import Any, Tuple

import numpy as np

from feinsum_amd.einsum import INT_CLASSES, Array, BatchedEinsum, ShapeComponentT, ShapeT, SizeParam


def _as_shape_component(s: Any) -> ShapeComponentT:
    if isinstance(s, str):
        return SizeParam(s)
    if isinstance(s, SizeParam) or (isinstance(s, INT_CLASSES) and s >= 0):
        return s
    raise ValueError(f"Cannot infer shape component '{s}'.")


def _as_shape(shape: Any) -> ShapeT:
    if isinstance(shape, str) or not isinstance(shape, Iterable):
        shape = (shape,)
    return tuple(_as_shape_component(d) for d in shape)


def array(name: str, shape: Any, dtype: Any = "float64") -> Array:
    """An operand named *name*; string axes are parametric (``"E"``)."""
    return Array(name=name, shape=_as_shape(shape), dtype=np.dtype(dtype))


def _parse_indices(subscript: str, is_output: bool) -> Tuple[str, ...]:
    indices = []
    rest = subscript.strip()
    while rest:
        if rest.startswith("..."):
            raise NotImplementedError("Broadcasting in einsums not supported")
        ch = rest[0]
        if not (ch.isascii() and ch.isalpha()):
            raise ValueError(f"Cannot parse '{rest}' in provided einsum '{subscript}'.")
        indices.append(ch)
        rest = rest[1:].lstrip()
    if is_output and len(set(indices)) != len(indices):
        raise ValueError(
            f"Used an input more than once to refer to the output axis in '{subscript}")
    return tuple(indices)


def batched_einsum(subscripts: str, args: Sequence[Sequence[Array]]) -> BatchedEinsum:
    """
    ``b`` einsums sharing *subscripts*; ``args[k]`` are the operands of the
    k-th one (interface of :func:`numpy.einsum`, explicit mode).
    """
    if "->" not in subscripts:
        raise ValueError("Missing -> in 'subscripts'. If the expected behavior"
                         " is implicit mode, feinsum does not support it.")
    in_specs, out_spec = subscripts.split("->")
    out_idx_set = _parse_indices(out_spec, is_output=True)
    in_idx_sets = tuple(_parse_indices(spec, is_output=False) for spec in in_specs.split(","))
    try:
        return BatchedEinsum(out_idx_set, in_idx_sets, tuple(tuple(row) for row in args))
    except AssertionError as exc:
        raise TypeError(f"{exc}") from exc


def einsum(subscripts: str, *operands: Array) -> BatchedEinsum:
    """A single einsum (``b == 1``)."""
    return batched_einsum(subscripts, [operands])
