"""
Recognise which hand-written kernel family a :class:`BatchedEinsum` belongs to.

This is the build's replacement for the reference's canonicalisation +
transform-archive lookup (reference: ``src/feinsum/canonicalization.py:1087``
``canonicalize_einsum`` and ``src/feinsum/sql_utils.py:160-294``
``query``/``retrieve``): instead of a graph canonical form keyed into sqlite, a
short table of templates is matched up to (a) renaming of indices and (b)
order of the operands.  Axis order inside an operand and inside the output is
memory layout and must match the template exactly.

Families: grad ``xre,rij,ej->xei``; div ``xre,rij,xej->ei``;
face-mass ``ef,fij,fej->ei`` with its layout siblings (J as ``fe``, operator as
``ifj`` -- ``tuning/impls/ifj_fe_fej_to_ei.py:46-60``), and the transposed-operator
siblings of all three (``rji``: ``tuning/impls/xre_rji_xej_to_ei_v1.py``; ``fji`` /
``jfi``: ``tuning/impls/jfi_fe_fej_to_ei.py:46-56``); the div component
``re,rij,ej->ei`` of ``test/test_codegen.py:34-66`` (batched over the three components);
the element-local operator ``e,ij,ej->ei`` / ``ij,ej->ei``
(``tuning/impls/e_ij_ej_to_ei_no_prftch.py``, ``ij_ej_to_ei_no_prftch.py``).
Anything else is evaluated by the generic einsum kernel.
"""

from __future__ import annotations

from dataclasses import dataclass
from itertools import permutations
from typing import Dict, Optional, Tuple

import numpy as np

from feinsum_amd.einsum import BatchedEinsum, SizeParam

FAMILY_GRAD, FAMILY_DIV, FAMILY_GRADDIV, FAMILY_FACEMASS, FAMILY_DIVCOMP, FAMILY_GRADPLANES = 1, 2, 3, 4, 5, 6
FAMILY_MATAPPLY = 7
FM_J_FE, FM_R_IFJ, FM_R_T = 1, 2, 4
OP_TRANSPOSED, OP_J_ES = 1, 2

# (family, layout_flags, subscripts, roles of the operands in template order)
_TEMPLATES = (
    (FAMILY_GRAD, 0, "xre,rij,ej->xei", ("J", "D", "u")),
    (FAMILY_GRAD, OP_TRANSPOSED, "xre,rji,ej->xei", ("J", "D", "u")),
    (FAMILY_DIV, 0, "xre,rij,xej->ei", ("J", "D", "u")),
    (FAMILY_DIV, OP_TRANSPOSED, "xre,rji,xej->ei", ("J", "D", "u")),     # xre_rji_xej_to_ei_v{0,1}.py
    # div components: test/test_codegen.py:34-66, re_rij_ej_to_ei.py, re_rji_ej_to_ei_*.py
    (FAMILY_DIVCOMP, 0, "re,rij,ej->ei", ("J", "D", "u")),
    (FAMILY_DIVCOMP, OP_TRANSPOSED, "re,rji,ej->ei", ("J", "D", "u")),
    (FAMILY_DIVCOMP, OP_J_ES, "er,rij,ej->ei", ("J", "D", "u")),        # examples/dg_wave_div.py
    (FAMILY_DIVCOMP, OP_J_ES | OP_TRANSPOSED, "er,rji,ej->ei", ("J", "D", "u")),
    # element-local operator with / without a per-element factor:
    # tuning/impls/e_ij_ej_to_ei_no_prftch.py:30-38, ij_ej_to_ei_no_prftch.py
    (FAMILY_MATAPPLY, 0, "e,ij,ej->ei", ("J", "D", "u")),
    (FAMILY_MATAPPLY, OP_TRANSPOSED, "e,ji,ej->ei", ("J", "D", "u")),
    (FAMILY_MATAPPLY, 0, "ij,ej->ei", ("D", "u")),
    (FAMILY_MATAPPLY, OP_TRANSPOSED, "ji,ej->ei", ("D", "u")),
) + tuple(
    (FAMILY_FACEMASS, jflag | rflag, f"{jsub},{rsub},fej->ei", ("J", "R", "v"))
    for jflag, jsub in ((0, "ef"), (FM_J_FE, "fe"))
    for rflag, rsub in ((0, "fij"), (FM_R_IFJ, "ifj"), (FM_R_T, "fji"), (FM_R_IFJ | FM_R_T, "jfi"))  # jfi_fe_fej_to_ei.py
)


@dataclass(frozen=True)
class KernelPlan:
    """How to evaluate an einsum with the HIP library.

    ``roles[k]`` maps a role (``"J"``, ``"D"``/``"R"``, ``"u"``/``"v"``) to the
    operand *position* in row k of ``einsum.args``; ``long_index`` is the einsum's
    own letter of the element axis; ``params`` holds Np / nf / Nfp.
    """

    family: int
    layout_flags: int
    roles: Dict[str, int]
    long_index: str
    params: Dict[str, int]

    @property
    def name(self) -> str:
        return {FAMILY_GRAD: "grad", FAMILY_DIV: "div", FAMILY_FACEMASS: "facemass",
                FAMILY_DIVCOMP: "divcomp", FAMILY_MATAPPLY: "matapply"}[self.family]


def _match_template(einsum: BatchedEinsum, subscripts: str) -> Optional[Tuple[Tuple[int, ...], Dict[str, str]]]:
    lhs, rhs = subscripts.split("->")
    t_in = [tuple(s) for s in lhs.split(",")]
    t_out = tuple(rhs)
    if len(t_in) != einsum.n or len(t_out) != len(einsum.out_idx_set):
        return None
    for perm in permutations(range(einsum.n)):
        # template operand k <-> einsum operand perm[k]
        mapping: Dict[str, str] = {}
        ok = True
        pairs = [(t_in[k], einsum.in_idx_sets[perm[k]]) for k in range(einsum.n)]
        pairs.append((t_out, einsum.out_idx_set))
        for t_idxs, e_idxs in pairs:
            if len(t_idxs) != len(e_idxs):
                ok = False
                break
            for t, e in zip(t_idxs, e_idxs):
                if mapping.setdefault(t, e) != e:
                    ok = False
                    break
            if not ok:
                break
        if ok and len(set(mapping.values())) == len(mapping) == len(einsum.all_indices):
            return perm, mapping
    return None


def match_family(einsum: BatchedEinsum) -> Optional[KernelPlan]:
    """Return the :class:`KernelPlan` for *einsum*, or ``None`` if it is not a DG-family einsum."""
    dtypes = {np.dtype(dt) for dt in einsum.arg_to_dtype.values()}
    if dtypes not in ({np.dtype("float64")}, {np.dtype("float32")}):
        return None     # mixed or other element types: the generic einsum kernel (or none)
    is_f32 = dtypes == {np.dtype("float32")}
    for family, flags, subscripts, roles in _TEMPLATES:
        m = _match_template(einsum, subscripts)
        if m is None:
            continue
        perm, mapping = m
        dim = lambda t: einsum.index_to_dim_length[mapping[t]]  # noqa: E731
        long_dim = dim("e")
        fixed = [t for t in mapping if t != "e"]
        if any(isinstance(dim(t), SizeParam) for t in fixed):
            continue
        if family in (FAMILY_GRAD, FAMILY_DIV):
            # tetrahedra (ndim = 3) and triangles (ndim = 2)
            if int(dim("x")) not in (2, 3) or int(dim("r")) != int(dim("x")) or int(dim("i")) != int(dim("j")):
                continue
            params = {"Np": int(dim("i")), "ndim": int(dim("x"))}
        elif family == FAMILY_DIVCOMP:
            if int(dim("r")) not in (2, 3) or int(dim("i")) != int(dim("j")):
                continue
            params = {"Np": int(dim("i")), "ndim": int(dim("r"))}
        elif family == FAMILY_MATAPPLY:
            if int(dim("i")) != int(dim("j")):
                continue
            params = {"Np": int(dim("i"))}
        else:
            params = {"Np": int(dim("i")), "nf": int(dim("f")), "Nfp": int(dim("j"))}
        del long_dim
        if is_f32:
            params["f32"] = 1      # all-float32 operands: fe_launch_f32 (the LDS-tiled kernel in float)
        return KernelPlan(family, flags, {role: perm[k] for k, role in enumerate(roles)},
                          mapping["e"], params)
    return None
