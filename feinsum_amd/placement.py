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

class _SplitBuffer:
    """Owner of one ``fe_split_alloc`` array; torch reads it through ``__cuda_array_interface__`` and keeps this
    object alive for as long as any tensor (or view) of the array lives; the memory returns to the pool with it."""

    def __init__(self, nbytes: int, device_index: int) -> None:
        from feinsum_amd import _hip

        self.ptr, self.nbytes, self.device_index = _hip.split_alloc(nbytes), int(nbytes), int(device_index)
        self._free = _hip.split_free     # (bound now: module globals may be gone at interpreter exit)
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (self.ptr, False),
                                         "version": 3, "strides": None}

    def __del__(self) -> None:
        ptr, self.ptr = getattr(self, "ptr", 0), 0
        if not ptr:
            return
        try:
            import torch

            with torch.cuda.device(self.device_index):
                self._free(ptr)          # waits for the device like hipFree, then unmaps
        except Exception:                # noqa: BLE001  (interpreter shutdown: the process is going away anyway)
            pass


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
        buf = _SplitBuffer(nbytes, dev.index)
        flat = torch.as_tensor(buf, device=dev)
    if flat.data_ptr() != buf.ptr:
        raise RuntimeError("torch copied the split allocator's array instead of wrapping it")
    return flat.view(dtype).view(shape)


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
