"""
Protocols of the public API (reference: ``src/feinsum/typing.py:35-45``).

``TransformT`` is what the reference's callers pass as ``transform=``: a callable
``(t_unit, insn_match=None, kernel_name=None) -> t_unit`` over a loopy translation
unit.  loopy does not exist in this build; such callables are accepted for
source compatibility and ignored (the hand-written kernel is selected by
:mod:`feinsum_amd.family`); a ``str`` / ``dict`` selects a kernel variant instead.
"""

from typing import Any, Mapping, Optional, Protocol, Union


class TransformT(Protocol):
    def __call__(self, t_unit: Any, insn_match: Optional[Any] = None,
                 kernel_name: Optional[str] = None) -> Any: ...


class ToStr(Protocol):
    def __str__(self) -> str: ...


#: what ``transform=`` accepts here: a reference-style callable (ignored), None, a variant name
#: ("auto" | "mfma" | "generic") or ``{"variant": name}``
VariantSelectorT = Union[TransformT, None, str, Mapping[str, Any]]
