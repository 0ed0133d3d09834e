"""
Device handles without a device (reference: ``src/feinsum/cl_utils.py:9-21``):
``FakeCLDevice(name)`` is enough wherever only ``device.name`` is needed, e.g.
``get_roofline_flop_rate(expr, FakeCLDevice("AMD Instinct MI355X").name)``.
"""

from dataclasses import dataclass
from typing import Protocol


class DeviceT(Protocol):
    """Anything with a ``name`` (``pyopencl.Device`` in the reference, ``DeviceQueue.device`` here)."""

    @property
    def name(self) -> str: ...


@dataclass(frozen=True, repr=True, eq=True)
class FakeCLDevice:
    name: str
