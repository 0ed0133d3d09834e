"""
Where the arrays a launch WRITES sit in device memory.

New functionality (the reference allocates every array separately through PyOpenCL,
``src/feinsum/measure.py:44-60,80-108``, and has no notion of placement).  On MI355X the same
launch on the same device runs 8-14 % apart depending on which classes of PHYSICAL memory its
output arrays lie in: write streams confined to one class reach 5.2 TB/s,
streams spread over two 6.8.  :func:`empty` / :func:`zeros` return ordinary torch tensors whose
4 MiB pieces alternate between two measured classes (``fe_split_alloc``); ``evaluate`` and
``timeit`` allocate their own outputs this way.  The kernels and their results do not depend on
placement.  (Round 2's arena scan -- ``tune_base`` / ``--placement tuned`` -- is gone: the
allocator replaced it.)
"""

from __future__ import annotations

from typing import Any, Dict, Sequence

MIB = 1 << 20


# --------------------------------------------------------------------------
# the split allocator (round 3): arrays whose 4 MiB pieces alternate between
# two classes of physical memory -- no arena, no timing scan
# (include/feinsum_hip.h, feinsum_amd/csrc/fe_split_alloc.h)
# --------------------------------------------------------------------------

# Arrays whose last tensor is gone are RECYCLED, not freed (round 5): an ``evaluate()`` that allocates its own
# outputs would otherwise pay a VMM map (0.5-2 ms) per call and, on release, a ``hipDeviceSynchronize()`` plus an unmap --
# against a 0.19 ms kernel -- and grow the process's reserved address space for ever (a range is never handed out twice:
# fe_split_alloc.h).  A released array keeps its mapping and waits, with an event recorded on the releasing thread's current
# stream, in a per-(device, size) list; the next ``empty`` of that size makes ITS current stream wait for the event (no host
# synchronisation) and takes the array as it is.  The same stream-ordering assumption as torch's caching allocator: work on
# other streams must have been ordered before the last tensor was dropped.  At most ``FEINSUM_SPLIT_RECYCLE_MIB`` (default
# 4096) MiB wait per device; beyond that the oldest arrays are really freed.
import logging as _logging
import os as _os
import threading as _threading
from collections import deque as _deque

_log = _logging.getLogger(__name__)
_recycle_lock = _threading.RLock()      # (re-entrant: a garbage collection inside a locked region may run another __del__)
_recycled: Dict[Any, Any] = {}          # (device index, nbytes) -> deque of (ptr, event)
_recycled_bytes: Dict[int, int] = {}    # device index -> bytes waiting
_recycle_stats = {"reused": 0, "freed": 0, "kept": 0, "free_failures": 0}


def _recycle_cap() -> int:
    return int(float(_os.environ.get("FEINSUM_SPLIT_RECYCLE_MIB", "4096")) * MIB)


def recycle_stats() -> Dict[str, int]:
    """Counters of the array recycling above: arrays ``reused`` / ``kept`` for reuse / really ``freed``, ``free_failures``,
    and the bytes waiting per device."""
    with _recycle_lock:
        return dict(_recycle_stats, waiting_bytes=dict(_recycled_bytes))


def recycle_trim(device_index: Any = None) -> int:
    """Really free the arrays waiting for reuse (all devices, or one); returns how many."""
    from feinsum_amd import _hip

    with _recycle_lock:
        keys = [k for k in _recycled if device_index is None or k[0] == device_index]
        victims = [(k[0], ptr) for k in keys for ptr, _ in _recycled.pop(k)]
        for k in keys:
            _recycled_bytes[k[0]] = 0
    n = 0
    for dev, ptr in victims:
        n += _really_free(_hip.split_free, dev, ptr)
    return n


def _really_free(free_fn: Any, device_index: int, ptr: int) -> int:
    try:
        import torch

        with torch.cuda.device(device_index):
            free_fn(ptr)             # waits for the device like hipFree, then unmaps
        _recycle_stats["freed"] += 1
        return 1
    except Exception as exc:         # noqa: BLE001  (a failed free is a leak: say so -- not silently)
        _recycle_stats["free_failures"] += 1
        try:
            _log.warning("fe_split_free(%#x) on device %d failed: %s -- the array's memory stays mapped", ptr, device_index, str(exc)[:200])
        except Exception:            # noqa: BLE001  (interpreter shutdown: logging may be gone)
            pass
        return 0


class _SplitBuffer:
    """Owner of one ``fe_split_alloc`` array; torch reads it through ``__cuda_array_interface__`` and keeps this
    object alive for as long as any tensor (or view) of the array lives; the array is recycled (see above) with it."""

    def __init__(self, nbytes: int, device_index: int) -> None:
        import torch

        from feinsum_amd import _hip

        self.nbytes, self.device_index = int(nbytes), int(device_index)
        self._free = _hip.split_free     # (bound now: module globals may be gone at interpreter exit)
        self.ptr = 0
        with _recycle_lock:
            waiting = _recycled.get((self.device_index, self.nbytes))
            if waiting:
                self.ptr, event = waiting.popleft()
                _recycled_bytes[self.device_index] -= self.nbytes
                _recycle_stats["reused"] += 1
            else:
                event = None
        if self.ptr:
            if event is not None:
                torch.cuda.current_stream(self.device_index).wait_event(event)   # stream-ordered: no host synchronisation
        else:
            self.ptr = _hip.split_alloc(nbytes)
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (self.ptr, False),
                                         "version": 3, "strides": None}

    def __del__(self) -> None:
        ptr, self.ptr = getattr(self, "ptr", 0), 0
        if not ptr:
            return
        victims = []
        try:
            import torch

            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("stream capture in progress")      # no event may be recorded now: free below (after the capture it syncs)
            event = torch.cuda.Event()
            event.record(torch.cuda.current_stream(self.device_index))
            with _recycle_lock:
                _recycled.setdefault((self.device_index, self.nbytes), _deque()).append((ptr, event))
                _recycled_bytes[self.device_index] = _recycled_bytes.get(self.device_index, 0) + self.nbytes
                _recycle_stats["kept"] += 1
                cap = _recycle_cap()
                while _recycled_bytes[self.device_index] > cap:     # the oldest arrays of the largest waiting size go
                    key = max((k for k in _recycled if k[0] == self.device_index and _recycled[k]), key=lambda k: k[1], default=None)
                    if key is None:
                        break
                    old_ptr, _ = _recycled[key].popleft()
                    _recycled_bytes[self.device_index] -= key[1]
                    victims.append(old_ptr)
            ptr = 0
        except Exception:                # noqa: BLE001  (interpreter shutdown, capture: free it now)
            pass
        for v in victims + ([ptr] if ptr else []):
            _really_free(self._free, self.device_index, v)


def empty(shape: Sequence[int], dtype: Any = None, device: Any = None, *, written: bool = True) -> Any:
    """
    A new uninitialised device tensor, as ``torch.empty`` -- for arrays a launch WRITES taken from the split allocator:
    the array's 4 MiB pieces alternate between two classes of physical memory, so every write stream of a launch is
    spread over both all the time.  That is what makes the DG launches run at 76-78 % of the HBM roofline instead of
    67-74 %: grad's three output planes, the four face-mass outputs and div's single output
    alike, with the default kernels.  No arena and no timing scan: the memory mapped is the array's size rounded up to
    2 MiB; arrays below 8 MiB, ``written=False`` and CPU devices get a plain ``torch.empty`` (where a read-only array
    lies does not matter).  The tensor is an ordinary torch tensor (views, copies, kernels); its memory returns to the
    allocator's pool when the last view is gone.
    """
    import torch

    dtype = dtype or torch.float64
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    shape = tuple(int(d) for d in shape)
    nbytes = int(torch.Size(shape).numel()) * torch.empty((), dtype=dtype).element_size()
    if not written or dev.type != "cuda" or nbytes < SPLIT_MIN_BYTES:
        return torch.empty(shape, dtype=dtype, device=dev)
    with torch.cuda.device(dev):
        try:
            buf = _SplitBuffer(nbytes, dev.index)
        except NotImplementedError as exc:
            # the allocator refuses an array whose pieces would all be of ONE class (no second class of physical memory within its
            # search budget: fe_split_alloc, FE_EUNSUPPORTED) -- the worst placement there is; an ordinary allocation is better
            # on average.  Counted: ``ordinary_fallbacks()``, and ``split_stats()["unsplit_refused"]``
            global _ordinary_fallbacks
            _ordinary_fallbacks += 1
            _log.info("split allocator: %s -- torch.empty for %d bytes", str(exc)[:120], nbytes)
            return torch.empty(shape, dtype=dtype, device=dev)
        flat = torch.as_tensor(buf, device=dev)
    if flat.data_ptr() != buf.ptr:
        raise RuntimeError("torch copied the split allocator's array instead of wrapping it")
    return flat.view(dtype).view(shape)


_ordinary_fallbacks = 0


def ordinary_fallbacks() -> int:
    """How many arrays of :func:`empty` this process took from torch because the split allocator refused them."""
    return _ordinary_fallbacks


def is_split(tensor: Any) -> bool:
    """Whether *tensor* lies in an array of the split allocator."""
    return bool(split_info(tensor))


def zeros(shape: Sequence[int], dtype: Any = None, device: Any = None, *, written: bool = True) -> Any:
    """:func:`empty`, zero-filled (what the reference's ``cla.zeros`` outputs are: ``src/feinsum/measure.py:44-60``)."""
    return empty(shape, dtype, device, written=written).zero_()


#: below two pieces of 4 MiB there is nothing to alternate (include/feinsum_hip.h)
SPLIT_MIN_BYTES = 8 * MIB


def split_info(tensor: Any) -> Dict[str, Any]:
    """What the allocator did for *tensor* (an array of :func:`empty`): ``{"bytes", "mapped_bytes", "piece_mib", "pieces",
    "pieces_by_class", "first_pieces": the classes of the first 16 pieces, "tail_bytes", "alloc_ms"}``; ``{}`` for any
    other tensor."""
    import torch

    from feinsum_amd import _hip
    from feinsum_amd.diagnostics import InvalidParameterError

    try:
        with torch.cuda.device(tensor.device):
            return _hip.split_info(tensor.untyped_storage().data_ptr())
    except (InvalidParameterError, RuntimeError):
        return {}


def split_stats(device: Any = None) -> Dict[str, Any]:
    """The pool of the split allocator on *device*: classes seen, free pieces, pieces created, probes, spacers, ms."""
    import torch

    from feinsum_amd import _hip

    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        return _hip.split_stats()


def split_reserve(nbytes: int, device: Any = None) -> None:
    """Announce the total size of the written arrays about to be allocated with :func:`empty` / :func:`zeros` on *device*
    (``fe_split_reserve``): the allocator collects half of it of each of two classes of physical memory at once, instead
    of array by array -- a class its search has left behind does not come back."""
    import torch

    from feinsum_amd import _hip

    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        _hip.split_reserve(int(nbytes))


def split_trim(device: Any = None) -> None:
    """Release the allocator's free pieces on *device* to the driver."""
    import torch

    from feinsum_amd import _hip

    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        _hip.split_trim()
