"""
``import feinsum as f`` -- the reference's import name on this backend.

north_star asks for a drop-in: code written against feinsum's builder and measure API
(``test/test_codegen.py:34-120``, ``test/test_measure.py:55-81``: ``f.array``, ``f.einsum``,
``f.batched_einsum``, ``f.timeit``, ``f.stringify_comparison_vs_roofline``, ``feinsum.measure``,
``feinsum.sql_utils`` ...) should run unmodified with this package on its path instead of the
reference.  Everything here is :mod:`feinsum_amd`; the reference's submodule names resolve to the
corresponding :mod:`feinsum_amd` modules.  (Do not install both: the names collide by design.)
"""

import sys as _sys

import feinsum_amd as _impl
from feinsum_amd import *  # noqa: F401,F403
from feinsum_amd import canonicalization, cl_utils, contraction_schedule, diagnostics, make_einsum, measure, sql_utils, typing

_einsum_mod = _sys.modules["feinsum_amd.einsum"]   # (the package attribute `einsum` is the builder function)

__all__ = _impl.__all__
__version__ = _impl.__version__

# `import feinsum.measure`, `from feinsum.einsum import BatchedEinsum`, ... (src/feinsum/*.py of the reference)
for _name, _mod in (("measure", measure), ("einsum", _einsum_mod), ("make_einsum", make_einsum),
                    ("contraction_schedule", contraction_schedule), ("diagnostics", diagnostics),
                    ("sql_utils", sql_utils), ("canonicalization", canonicalization), ("cl_utils", cl_utils),
                    ("typing", typing)):
    _sys.modules[f"{__name__}.{_name}"] = _mod
del _sys, _name, _mod
