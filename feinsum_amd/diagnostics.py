"""
Exception types of the evaluator; same names and base classes as the
reference's ``feinsum.diagnostics`` (reference: ``src/feinsum/diagnostics.py:33-65``)
so that ``except`` clauses written against feinsum keep working.
"""


class EinsumTunitMatchError(ValueError):
    """An einsum could not be matched against a kernel family / pattern."""


class InvalidParameterError(ValueError):
    """A kernel parameter (shape, variant, pointer) is in range but illegal.

    Also raised for ``FE_EINVAL`` returned by the HIP library.
    """


class NoDevicePeaksInfoError(LookupError):
    """No peak flop-rate / bandwidth entry for the queried device."""


class TransformValidationError(RuntimeError):
    """The device result does not reproduce ``numpy.einsum`` within tolerance."""


class NoFactInDatabaseError(RuntimeError):
    """Kept for API compatibility: the build has no transform archive."""


class HipLibraryError(RuntimeError):
    """libfeinsum_hip.so is missing, fails to load, or reported a HIP error."""
