"""
Evaluate, validate and time a :class:`BatchedEinsum` on an MI355X.

API mirror of the reference's ``feinsum.measure`` (reference:
``src/feinsum/measure.py``): :func:`timeit` (``:197-275``),
:func:`validate_batched_einsum_transform` (``:111-194``),
:func:`measure_giga_op_rate` (``:357-385``), :func:`get_roofline_flop_rate`
(``:388-418``), :func:`stringify_comparison_vs_roofline` (``:484-525``), the
input generator (``:63-108``) and the three protocol constants (``:35-37``).

What is different underneath: the reference lowers the einsum through loopy to
OpenCL and enqueues it with PyOpenCL; here :func:`evaluate` picks a
hand-written gfx950 kernel (``feinsum_amd.family``) and launches it through the
C ABI of ``libfeinsum_hip.so`` on a HIP stream.  PyTorch-ROCm is used only to
own device memory and streams.

Argument conventions kept from the reference:

``cq``         anything that names a device/stream: ``None`` (device 0), an int
               ordinal, a ``torch.device``, or a :class:`DeviceQueue`.
``transform``  the reference passes a loopy transformation here; loopy does not
               exist in this build, so a callable (or ``None``) is ignored and
               selects the default kernel variant, while a ``str`` / ``dict``
               (``"mfma"``, ``"generic"``, ``{"variant": "generic"}``) selects
               a variant explicitly; in a dict, ``"prepared": True`` lets a
               bound launch (``timeit``) use a prepared copy of its operator
               matrices; ``"placement"`` (or ``$FEINSUM_PLACEMENT``): ``timeit``
               allocates one array per operand as the reference does; with the
               default ``"split"`` the outputs come from the split allocator
               (``feinsum_amd.placement.zeros``; ``evaluate`` allocates the outputs it
               is not handed the same way), ``"separate"`` takes every array
               from torch; ``timeit_details(...).placement`` reports which was used.
``schedule``   accepted and ignored: the kernels implement the optimal schedule.

Inputs are drawn from ``numpy.random.default_rng(0)`` in **sorted argument-name
order** (the reference draws in hash order, which is not reproducible).
"""

from __future__ import annotations

import logging
from ctypes import byref as C_byref
from dataclasses import dataclass, field
from time import time
from types import MappingProxyType
from typing import Any, Dict, Mapping, Optional, Sequence, Tuple

import numpy as np

from feinsum_amd import _hip
from feinsum_amd.contraction_schedule import ContractionSchedule, count_ops
from feinsum_amd.diagnostics import (HipLibraryError, InvalidParameterError,
                                     NoDevicePeaksInfoError, TransformValidationError)
from feinsum_amd.einsum import INT_CLASSES, BatchedEinsum, SizeParam
from feinsum_amd.family import (FAMILY_DIV, FAMILY_DIVCOMP, FAMILY_FACEMASS, FAMILY_GRAD,
                                FAMILY_GRADPLANES, FAMILY_MATAPPLY, OP_J_ES, KernelPlan,
                                match_family)

logger = logging.getLogger(__name__)

N_WARMUP_ROUNDS = 5
N_MIN_TIMING_ROUNDS = 10
N_MIN_SIM_SECS = 2
LAUNCHES_PER_BATCH = 5


# --------------------------------------------------------------------------
# device / stream handle (stands in for pyopencl.CommandQueue)
# --------------------------------------------------------------------------

@dataclass(frozen=True)
class DeviceInfo:
    name: str


class DeviceQueue:
    """A HIP device + stream; duck-types the bits of ``cl.CommandQueue`` the
    reference uses (``cq.device.name``, ``cq.finish()``)."""

    def __init__(self, device: Any = 0, stream: Any = None) -> None:
        import torch

        if not torch.cuda.is_available():
            raise HipLibraryError("no HIP device visible to this process")
        self.torch_device = torch.device("cuda", device) if isinstance(device, INT_CLASSES) \
            else torch.device(device)
        if self.torch_device.index is None:
            self.torch_device = torch.device("cuda", torch.cuda.current_device())
        self._stream = stream

    @property
    def ordinal(self) -> int:
        return int(self.torch_device.index)

    @property
    def stream(self):
        import torch

        return self._stream if self._stream is not None else torch.cuda.current_stream(self.torch_device)

    @property
    def stream_ptr(self) -> int:
        return int(self.stream.cuda_stream)

    @property
    def device(self) -> DeviceInfo:
        name, _, _ = _hip.device_info(self.ordinal)
        return DeviceInfo(name)

    def finish(self) -> None:
        self.stream.synchronize()


def _as_queue(cq: Any) -> DeviceQueue:
    if isinstance(cq, DeviceQueue):
        return cq
    if cq is None:
        return DeviceQueue(0)
    return DeviceQueue(cq)


def _variant_from_transform(transform: Any):
    if transform is None or callable(transform):
        return None
    if isinstance(transform, Mapping):
        return transform.get("variant")
    return transform


def _prepared_from_transform(transform: Any, default: bool) -> bool:
    """``{"prepared": False}`` in a transform dict turns prepared operators off for a bound launch."""
    if isinstance(transform, Mapping) and "prepared" in transform:
        return bool(transform["prepared"])
    return default


# --------------------------------------------------------------------------
# arrays
# --------------------------------------------------------------------------

def get_real_dtype(dtype: np.dtype) -> np.dtype:
    return np.empty(0, dtype=dtype).real.dtype


def _concrete_shape(shape: Sequence[Any], long_dim_length: int) -> Tuple[int, ...]:
    return tuple(int(d) if isinstance(d, INT_CLASSES) else int(long_dim_length) for d in shape)


def _random_array(rng: np.random.Generator, dtype: np.dtype, shape: Tuple[int, ...]) -> np.ndarray:
    # reference: measure.py:63-77
    if dtype.kind == "c":
        real = get_real_dtype(dtype)
        return (rng.random(size=shape, dtype=real) + dtype.type(1j) * rng.random(size=shape, dtype=real))
    if dtype.kind == "i":
        return rng.integers(low=-100, high=100, size=shape, dtype=dtype)
    return rng.random(size=shape, dtype=dtype)


def generate_host_input_arrays(einsum: BatchedEinsum, long_dim_length: int,
                               np_seed: int = 0) -> Dict[str, np.ndarray]:
    """Host inputs: one ``default_rng(np_seed)``, arrays drawn in sorted-name order."""
    rng = np.random.default_rng(np_seed)
    return {name: _random_array(rng, einsum.arg_to_dtype[name],
                                _concrete_shape(einsum.arg_to_shape[name], long_dim_length))
            for name in sorted(einsum.arg_to_dtype)}


def generate_input_arrays(cq: Any, einsum: BatchedEinsum, long_dim_length: int,
                          np_seed: int = 0) -> Mapping[str, Any]:
    """Device inputs (reference: measure.py:80-108)."""
    import torch

    q = _as_queue(cq)
    host = generate_host_input_arrays(einsum, long_dim_length, np_seed)
    return MappingProxyType({name: torch.from_numpy(arr).to(q.torch_device)
                             for name, arr in host.items()})


def result_dtype(einsum: BatchedEinsum, row: int = 0) -> np.dtype:
    """dtype of an output = ``np.result_type`` of its operands (codegen/loopy.py:258-260)."""
    return np.result_type(*[arg.dtype for arg in einsum.args[row]])


def generate_out_arrays(cq: Any, einsum: BatchedEinsum, long_dim_length: int, *, split: bool = False) -> Mapping[str, Any]:
    """Zero-filled device outputs ``_fe_out, _fe_out_0, ...`` (reference: measure.py:44-60).  *split*: one array each
    from the split allocator (``feinsum_amd.placement.zeros``: 4 MiB pieces alternating between two classes of physical
    memory) instead of the torch allocator; arrays below 8 MiB come from torch either way."""
    import torch

    q = _as_queue(cq)
    shape = _concrete_shape(einsum.shape, long_dim_length)
    outs = {}
    if split and len(einsum.output_names) > 1:   # several arrays, allocated one after the other: say what is coming
        from feinsum_amd import placement

        nbytes = sum(int(np.prod(shape)) * result_dtype(einsum, k).itemsize for k in range(len(einsum.output_names)))
        if nbytes >= placement.SPLIT_MIN_BYTES:
            placement.split_reserve(nbytes, q.torch_device)
    for k, name in enumerate(einsum.output_names):
        tdtype = getattr(torch, result_dtype(einsum, k).name)
        if split:
            from feinsum_amd import placement

            outs[name] = placement.zeros(shape, tdtype, q.torch_device)
        else:
            outs[name] = torch.zeros(shape, dtype=tdtype, device=q.torch_device)
    return MappingProxyType(outs)


# --------------------------------------------------------------------------
# the waist: run one BatchedEinsum on device arrays
# --------------------------------------------------------------------------

def _check_tensor(name: str, t: Any, shape: Tuple[int, ...], dtype: np.dtype, q: DeviceQueue) -> None:
    import torch

    if not isinstance(t, torch.Tensor):
        raise InvalidParameterError(f"argument '{name}' must be a torch.Tensor on the device")
    if t.device != q.torch_device:
        raise InvalidParameterError(f"argument '{name}' lives on {t.device}, expected {q.torch_device}")
    if tuple(t.shape) != tuple(shape):
        raise InvalidParameterError(f"argument '{name}' has shape {tuple(t.shape)}, expected {shape}")
    if t.dtype != getattr(torch, dtype.name):
        raise InvalidParameterError(f"argument '{name}' has dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise InvalidParameterError(f"argument '{name}' must be C-contiguous")


def _long_length(einsum: BatchedEinsum, arg_dict: Mapping[str, Any]) -> Dict[str, int]:
    """Values of the size parameters, read off the arrays."""
    values: Dict[str, int] = {}
    for name, shape in einsum.arg_to_shape.items():
        for axis, d in enumerate(shape):
            if isinstance(d, SizeParam):
                got = int(arg_dict[name].shape[axis])
                if values.setdefault(d.name, got) != got:
                    raise InvalidParameterError(f"inconsistent values for size parameter '{d.name}'")
    return values


class _FamilyLaunch:
    """A family plan bound to concrete device arrays: launch / time."""

    def __init__(self, plan: KernelPlan, einsum: BatchedEinsum, arg_dict: Mapping[str, Any],
                 outs: Sequence[Any], variant: Any) -> None:
        self.plan, self.variant = plan, _hip.variant_code(variant)
        role = plan.roles
        rows = einsum.args
        first = rows[0]
        long_role = "J" if "J" in role else "u"       # 'ij,ej->ei' has no geometry factor
        long_axis = einsum.in_idx_sets[role[long_role]].index(plan.long_index)
        self.E = int(arg_dict[first[role[long_role]].name].shape[long_axis])
        self._keep = (arg_dict, outs)
        self._prepared: Dict[Any, Any] = {}   # (operator pointer, flags) -> prepared device buffer
        p = plan.params
        self.f32 = bool(p.get("f32"))         # all-float32 einsum: fe_launch_f32, whatever the variant
        op_role = "D" if plan.family in (FAMILY_GRAD, FAMILY_DIV, FAMILY_DIVCOMP, FAMILY_MATAPPLY) else "R"
        self.groups = []   # list of ArgPack (one per launch)
        in_role = "u" if op_role == "D" else "v"
        self.group_family = plan.family
        if plan.family == FAMILY_DIVCOMP and not self.f32 and self._bind_planes(einsum, arg_dict, outs):
            return
        # consecutive rows sharing J and the operator become one multi-field launch
        k = 0
        while k < len(rows):
            jname = lambda row: row[role["J"]].name if "J" in role else None   # noqa: E731
            jn, rn = jname(rows[k]), rows[k][role[op_role]].name
            k2 = k + 1
            while (plan.family != FAMILY_DIVCOMP and k2 < len(rows)
                   and jname(rows[k2]) == jn and rows[k2][role[op_role]].name == rn):
                k2 += 1
            vptrs = [arg_dict[rows[m][role[in_role]].name].data_ptr() for m in range(k, k2)]
            optrs = [outs[m].data_ptr() for m in range(k, k2)]
            pack = _hip.ArgPack()
            pack.J = arg_dict[jn].data_ptr() if jn is not None else None
            pack.D = arg_dict[rn].data_ptr()
            pack.u, pack.out = vptrs[0], optrs[0]
            va, oa = _hip._ptr_array(vptrs), _hip._ptr_array(optrs)
            self._keep += (va, oa)
            pack.v, pack.outs = va, oa
            pack.E, pack.Np = self.E, p["Np"]
            pack.nf, pack.Nfp, pack.ndim = p.get("nf", 0), p.get("Nfp", 0), p.get("ndim", 3)
            pack.b, pack.layout_flags, pack.variant = k2 - k, plan.layout_flags, self.variant
            self.groups.append(pack)
            k = k2

    def __del__(self) -> None:
        # the library keeps a record per prepared buffer (address -> shape, source operator): drop it before the buffer
        # is freed, or an unrelated later allocation at the same address would be taken for a prepared operator
        for buf in getattr(self, "_prepared", {}).values():
            try:
                _hip.release_prepared(buf.data_ptr())
            except Exception:   # noqa: BLE001  (interpreter shutdown)
                pass

    def prepare_operators(self, stream_ptr: int = 0) -> int:
        """Write the operator of every launch group that has a prepared form (grad / div / face-mass of
        tetrahedra p = 1..4, ``fe_prepare_operator``) into a device buffer owned by this bound launch;
        the launches then fetch their MFMA fragments from it instead of rebuilding them from the
        plain array every time.  The buffers are SNAPSHOTS: call this again after changing an
        operator array in place.  Returns the number of prepared groups."""
        import torch

        if self.f32 or self.variant not in (_hip.VARIANT_AUTO, _hip.VARIANT_MFMA, _hip.VARIANT_MFMA_SPLIT) or self.group_family == FAMILY_GRADPLANES \
                or self.plan.family not in (FAMILY_GRAD, FAMILY_DIV, FAMILY_FACEMASS):
            return 0
        done = 0
        for pack in self.groups:
            if self.plan.family != FAMILY_FACEMASS and pack.ndim != 3:
                continue
            key = (pack.D, pack.layout_flags)
            buf = self._prepared.get(key)
            fresh = buf is None
            if fresh:
                device = self._keep[1][0].device
                buf = torch.empty(_hip.PREPARED_OPERATOR_BYTES, dtype=torch.uint8, device=device)
            flags = pack.layout_flags & ~1 if self.plan.family == FAMILY_FACEMASS else pack.layout_flags   # (not FM_J_FE)
            try:
                _hip.prepare_operator(self.plan.family, pack.D, pack.Np, pack.nf, pack.Nfp, flags, buf.data_ptr(),
                                      stream_ptr)
            except NotImplementedError:
                continue
            self._prepared[key] = buf
            pack.prepared = buf.data_ptr()
            done += 1
        return done

    def _bind_planes(self, einsum: BatchedEinsum, arg_dict: Mapping[str, Any], outs: Sequence[Any]) -> bool:
        """Rows of 're,rij,ej->ei' that share u and D (e.g. the curl-type batch of
        ``tuning/impls/re_rji_ej_to_ei_3d_cross_product_v0.py:220-231``) go through the grad-type
        planes launch, which forms D u once per field.  False: keep one launch per row."""
        role, rows = self.plan.roles, einsum.args
        if self.plan.layout_flags & OP_J_ES or self.plan.params.get("ndim", 3) != 3:
            return False
        if len({row[role["D"]].name for row in rows}) != 1:
            return False
        jnames = sorted({row[role["J"]].name for row in rows})
        fields: dict = {}
        for m, row in enumerate(rows):
            fields.setdefault(row[role["u"]].name, {})[row[role["J"]].name] = m
        n_planes = {len(planes) for planes in fields.values()}
        if (len(jnames) > 3 or len(n_planes) != 1 or next(iter(n_planes)) < 2
                or sum(len(planes) for planes in fields.values()) != len(rows)):
            return False
        jptrs = [arg_dict[jnames[min(x, len(jnames) - 1)]].data_ptr() for x in range(3)]
        uptrs = [arg_dict[name].data_ptr() for name in fields]
        optrs = [outs[planes[jn]].data_ptr() if jn in planes else None
                 for planes in fields.values() for jn in (jnames + [None] * 3)[:3]]
        pack = _hip.ArgPack()
        ja, ua, oa = _hip._ptr_array(jptrs), _hip._ptr_array(uptrs), _hip._ptr_array(optrs)
        self._keep += (ja, ua, oa)
        pack.j3, pack.v, pack.outs = ja, ua, oa
        pack.D = arg_dict[rows[0][role["D"]].name].data_ptr()
        pack.E, pack.Np, pack.b = self.E, self.plan.params["Np"], len(uptrs)
        pack.layout_flags, pack.variant = self.plan.layout_flags, self.variant
        self.groups.append(pack)
        self.group_family = FAMILY_GRADPLANES
        return True

    def launch(self, stream_ptr: int) -> None:
        lib = _hip.load_library()
        for pack in self.groups:
            if self.f32:
                _hip.check(lib.fe_launch_f32(self.plan.family, C_byref(pack), stream_ptr))
            elif self.group_family == FAMILY_GRADPLANES:
                _hip.check(lib.fe_gradplanes3d_f64(pack.j3, pack.D, pack.v, pack.outs, pack.E, pack.Np,
                                                   pack.b, pack.layout_flags, pack.variant, stream_ptr))
            elif self.plan.family == FAMILY_GRAD and pack.prepared:
                _hip.check(lib.fe_grad3d_prepared_f64(pack.J, pack.D, pack.prepared, pack.v, pack.outs, pack.E,
                                                      pack.Np, pack.b, pack.layout_flags, pack.variant, stream_ptr))
            elif self.plan.family == FAMILY_GRAD:
                _hip.check(lib.fe_grad_f64(pack.J, pack.D, pack.v, pack.outs, pack.E, pack.ndim, pack.Np,
                                           pack.b, pack.layout_flags, pack.variant, stream_ptr))
            elif self.plan.family == FAMILY_DIV and pack.prepared:
                _hip.check(lib.fe_div3d_prepared_f64(pack.J, pack.D, pack.prepared, pack.v, pack.outs, pack.E,
                                                     pack.Np, pack.b, pack.layout_flags, pack.variant, stream_ptr))
            elif self.plan.family == FAMILY_DIV:
                _hip.check(lib.fe_div_f64(pack.J, pack.D, pack.v, pack.outs, pack.E, pack.ndim, pack.Np,
                                          pack.b, pack.layout_flags, pack.variant, stream_ptr))
            elif self.plan.family == FAMILY_MATAPPLY:
                _hip.check(lib.fe_matapply_f64(pack.J, pack.D, pack.v, pack.outs, pack.E, pack.Np, pack.b,
                                               pack.layout_flags, pack.variant, stream_ptr))
            elif self.plan.family == FAMILY_DIVCOMP:
                _hip.check(lib.fe_divcomp_f64(pack.J, pack.D, pack.u, pack.out, pack.E, pack.ndim, pack.Np,
                                              pack.layout_flags, pack.variant, stream_ptr))
            else:
                _hip.check(lib.fe_facemass_prepared_f64(pack.J, pack.D, pack.prepared, pack.v, pack.outs, pack.E,
                                                        pack.Np, pack.nf, pack.Nfp, pack.b, pack.layout_flags,
                                                        pack.variant, stream_ptr))

    def time_batch(self, n: int, stream_ptr: int) -> float:
        """Seconds for *n* launches of the whole batched einsum (HIP events)."""
        if len(self.groups) == 1:
            family = self.group_family | (_hip.FAMILY_F32 if self.f32 else 0)
            return _hip.time_launches(family, self.groups[0], n, stream_ptr) * 1e-3
        import torch

        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream = torch.cuda.ExternalStream(stream_ptr) if stream_ptr else torch.cuda.current_stream()
        t0.record(stream)
        for _ in range(n):
            self.launch(stream_ptr)
        t1.record(stream)
        t1.synchronize()
        return t0.elapsed_time(t1) * 1e-3


class _GenericLaunch:
    """Any other einsum: one generic-kernel launch per output row."""

    def __init__(self, einsum: BatchedEinsum, arg_dict: Mapping[str, Any], outs: Sequence[Any]) -> None:
        dtypes = {np.dtype(dt) for dt in einsum.arg_to_dtype.values()}
        if len(dtypes) != 1 or next(iter(dtypes)) not in (np.dtype("float64"), np.dtype("float32")):
            raise NotImplementedError(
                "the generic einsum kernel is compiled for all-float64 or all-float32 operands;"
                f" got {sorted(str(d) for d in dtypes)}")
        dtype = next(iter(dtypes))
        sizes = _long_length(einsum, arg_dict)
        extent = {idx: (sizes[d.name] if isinstance(d, SizeParam) else int(d))
                  for idx, d in einsum.index_to_dim_length.items()}
        if einsum.n > _hip.FE_MAX_EINSUM_OPERANDS or len(einsum.out_idx_set) > _hip.FE_MAX_EINSUM_INDICES \
                or len(einsum.sum_indices) > _hip.FE_MAX_EINSUM_INDICES:
            raise NotImplementedError("einsum has more operands / indices than the generic kernel supports")
        self._keep = (arg_dict, outs)
        self.launches = []
        for row, out in zip(einsum.args, outs):
            d = _hip.EinsumDesc()
            d.n_operands, d.n_out, d.n_sum = einsum.n, len(einsum.out_idx_set), len(einsum.sum_indices)
            d.dtype = 0 if dtype == np.dtype("float64") else 1
            for k, idx in enumerate(einsum.out_idx_set):
                d.out_extent[k] = extent[idx]
            for k, idx in enumerate(einsum.sum_indices):
                d.sum_extent[k] = extent[idx]
            for p, (arg, idxs) in enumerate(zip(row, einsum.in_idx_sets)):
                strides = arg_dict[arg.name].stride()
                for axis, idx in enumerate(idxs):
                    if idx in einsum.out_idx_set:
                        d.op_out_stride[p][einsum.out_idx_set.index(idx)] += strides[axis]
                    else:
                        d.op_sum_stride[p][einsum.sum_indices.index(idx)] += strides[axis]
            self.launches.append((d, [arg_dict[a.name].data_ptr() for a in row], out.data_ptr()))

    def launch(self, stream_ptr: int) -> None:
        for d, ops, out in self.launches:
            _hip.einsum_generic(d, ops, out, stream_ptr)

    def time_batch(self, n: int, stream_ptr: int) -> float:
        import torch

        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream = torch.cuda.ExternalStream(stream_ptr) if stream_ptr else torch.cuda.current_stream()
        t0.record(stream)
        for _ in range(n):
            self.launch(stream_ptr)
        t1.record(stream)
        t1.synchronize()
        return t0.elapsed_time(t1) * 1e-3


def _bind(einsum: BatchedEinsum, cq: Any, arg_dict: Mapping[str, Any],
          out_dict: Optional[Mapping[str, Any]], transform: Any, prepare: bool = False):
    import torch

    q = _as_queue(cq)
    missing = sorted(set(einsum.all_args) - set(arg_dict))
    if missing:
        raise InvalidParameterError(f"missing input arrays: {missing}")
    sizes = _long_length(einsum, arg_dict)
    for name, shape in einsum.arg_to_shape.items():
        concrete = tuple(sizes[d.name] if isinstance(d, SizeParam) else int(d) for d in shape)
        _check_tensor(name, arg_dict[name], concrete, einsum.arg_to_dtype[name], q)
    out_shape = tuple(sizes[d.name] if isinstance(d, SizeParam) else int(d) for d in einsum.shape)
    outs = []
    allocated: dict = {}      # outputs this call allocated itself: name -> "split" | "torch" | "torch (<why>)"
    for k, name in enumerate(einsum.output_names):
        dt = result_dtype(einsum, k)
        if out_dict is not None and name in out_dict:
            _check_tensor(name, out_dict[name], out_shape, dt, q)
            outs.append(out_dict[name])
        else:
            tensor, how = _allocate_output(out_shape, getattr(torch, dt.name), q.torch_device, transform)
            outs.append(tensor)
            allocated[name] = how
    plan = match_family(einsum)
    variant = _variant_from_transform(transform)
    if plan is not None:
        bound = _FamilyLaunch(plan, einsum, arg_dict, outs, variant)
        if _prepared_from_transform(transform, prepare):
            with torch.cuda.device(q.torch_device):
                bound.prepare_operators(q.stream_ptr)
    else:
        if variant not in (None, "auto", "generic", 0, 1):
            raise NotImplementedError(
                f"einsum '{einsum.get_subscripts()}' is outside the DG kernel families;"
                " only the generic kernel is available for it")
        bound = _GenericLaunch(einsum, arg_dict, outs)
    # byte ranges the launch reads and writes (operator.py checks them before reordering launches)
    span = lambda t: (int(t.data_ptr()), int(t.numel()) * int(t.element_size()))   # noqa: E731
    bound.reads = tuple(span(arg_dict[name]) for name in sorted(einsum.all_args))
    bound.writes = tuple(span(t) for t in outs)
    bound.output_allocations = MappingProxyType(allocated)
    return q, bound, outs


def _allocate_output(shape: Tuple[int, ...], dtype: Any, device: Any, transform: Any):
    """
    An output array the caller did not supply (the reference allocates them through its PyOpenCL pool,
    ``src/feinsum/measure.py:44-60,236-246``).  Arrays of 8 MiB and more come from the split allocator
    (``placement.empty``: on MI355X a launch writing into its arrays runs 5-12 % faster than into ordinary allocations,
    DESIGN.md section 3d) unless ``transform={"placement": "separate"}`` / ``$FEINSUM_PLACEMENT=separate`` asks for plain
    ones; whatever the allocator cannot serve (no VMM support, address-space cap, out of memory in its pool) is a plain
    ``torch.empty``.  Returns ``(tensor, how)``.
    """
    import torch

    from feinsum_amd import placement

    nbytes = int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dtype).element_size() if shape else 0
    if torch.device(device).type != "cuda" or nbytes < placement.SPLIT_MIN_BYTES:
        return torch.empty(shape, dtype=dtype, device=device), "torch"
    if _placement_mode(transform) != "split":
        return torch.empty(shape, dtype=dtype, device=device), "torch (placement: separate)"
    if torch.cuda.is_current_stream_capturing():
        # the allocator creates handles, probes with kernels of its own and may synchronise the device: none of that may
        # happen inside a stream capture -- torch's allocator knows how to allocate for a graph
        return torch.empty(shape, dtype=dtype, device=device), "torch (stream capture in progress)"
    try:
        t = placement.empty(shape, dtype, device)
        # (the allocator refuses an array that would be of one class of physical memory: placement.empty then allocates ordinarily)
        return t, "split" if placement.is_split(t) else "torch (the split allocator found no second class of physical memory)"
    except (RuntimeError, HipLibraryError) as exc:
        logger.warning("split allocator not available (%s); the output is an ordinary allocation", str(exc)[:160])
        return torch.empty(shape, dtype=dtype, device=device), f"torch (split allocator failed: {str(exc)[:120]})"


def evaluate(einsum: BatchedEinsum, cq: Any, arg_dict: Mapping[str, Any], *,
             out_dict: Optional[Mapping[str, Any]] = None, transform: Any = None,
             wait: bool = False) -> Mapping[str, Any]:
    """
    Enqueue *einsum* on the queue's stream and return ``{"_fe_out": tensor, ...}``
    (the replacement for ``t_unit.executor(cq, ...)(cq, **arg_dict)``,
    reference measure.py:163-165).  Asynchronous unless *wait*; outputs are
    fully overwritten.
    """
    import torch

    q, bound, outs = _bind(einsum, cq, arg_dict, out_dict, transform)
    with torch.cuda.device(q.torch_device):
        bound.launch(q.stream_ptr)
    if wait:
        q.finish()
    return MappingProxyType(dict(zip(einsum.output_names, outs)))


# --------------------------------------------------------------------------
# validation and timing
# --------------------------------------------------------------------------

def _tolerances(dtype: np.dtype) -> Tuple[float, float]:
    real = get_real_dtype(np.dtype(dtype))
    if real == np.float32:
        return 1e-6, 1e-6
    if real == np.float64:
        return 1e-10, 1e-10
    raise NotImplementedError(real)


def validate_batched_einsum_transform(einsum: BatchedEinsum, cq: Any, transform: Any,
                                      schedule: Optional[ContractionSchedule] = None) -> None:
    """
    Run the selected kernel at ``long_dim_length = 100`` and compare every output
    with ``np.einsum(subscripts, *inputs, optimize="optimal")``; atol = rtol =
    1e-10 (float64) / 1e-6 (float32).  Raises
    :class:`~feinsum_amd.diagnostics.TransformValidationError` on mismatch.
    (reference: measure.py:111-194)
    """
    del schedule
    long_dim_length = 100
    q = _as_queue(cq)
    host = generate_host_input_arrays(einsum, long_dim_length)
    import torch

    arg_dict = {name: torch.from_numpy(arr).to(q.torch_device) for name, arr in host.items()}
    ref_outs = {name: np.einsum(einsum.get_subscripts(), *[host[arg.name] for arg in row],
                                optimize="optimal")
                for name, row in zip(einsum.output_names, einsum.args)}
    outs = evaluate(einsum, q, arg_dict, transform=transform, wait=True)
    if set(ref_outs) != set(outs):
        raise RuntimeError("Output names mismatch")
    for name in sorted(ref_outs):
        got = outs[name].cpu().numpy()
        ref = ref_outs[name]
        if got.dtype != ref.dtype:
            raise RuntimeError(f"dtype mismatch for output '{name}'")
        atol, rtol = _tolerances(ref.dtype)
        try:
            np.testing.assert_allclose(got, ref, atol=atol, rtol=rtol)
        except AssertionError as exc:
            raise TransformValidationError(f"{exc}") from exc
    logger.info("Statistically verified the soundness of the transformation")


PLACEMENT_MODES = ("split", "separate")


def _placement_mode(transform: Any) -> str:
    """
    Where ``timeit`` puts the arrays it allocates: ``transform={"placement": ...}``, else ``$FEINSUM_PLACEMENT``, else
    ``"split"``.

    ``"split"``     one allocation per array, as the reference does (``src/feinsum/measure.py:44-60,80-108``); the
                    OUTPUTS come from the split allocator (``feinsum_amd.placement.zeros``), whose arrays alternate
                    between two classes of physical memory every 4 MiB -- no arena, no scan, memory = the footprint.
                    ``evaluate`` allocates the outputs it is not handed the same way, and a caller gets such arrays
                    with ``placement.empty``.
    ``"separate"``  every array from the torch allocator: the reference's protocol to the letter, and what a caller
                    who allocates with ``torch.empty`` gets.
    The :class:`TimingResult` says which one was used (``record_facts`` stores it with the fact).
    """
    import os

    mode = transform.get("placement") if isinstance(transform, Mapping) else None
    mode = mode or os.environ.get("FEINSUM_PLACEMENT") or "split"
    if mode == "auto":          # round 2's default name
        mode = "split"
    if mode not in PLACEMENT_MODES:
        raise InvalidParameterError(f"placement must be one of {PLACEMENT_MODES}, got {mode!r}")
    return mode


@dataclass(frozen=True)
class TimingResult:
    """What :func:`timeit_details` measured (seconds are per launch)."""

    seconds_device: float     # HIP-event time per launch (what timeit returns)
    seconds_wall: float       # host wall-clock per launch, reference protocol
    rounds: int
    #: how the timed arrays were placed: {"mode": "split", "outputs": the allocator's report per output} (one allocation
    #: per array, outputs from the split allocator) or {"mode": "separate"} (every array from torch); a placement that
    #: could not be had says so under "fallback"
    placement: Mapping[str, Any] = field(default_factory=lambda: MappingProxyType({"mode": "separate"}))


def timeit_details(einsum: BatchedEinsum, *, transform: Any = None, cq: Any = None,
                   long_dim_length: int = 100000,
                   schedule: Optional[ContractionSchedule] = None,
                   min_rounds: int = N_MIN_TIMING_ROUNDS,
                   min_secs: float = N_MIN_SIM_SECS, validate: bool = True) -> TimingResult:
    """The reference's timing protocol (measure.py:248-275) with both clocks reported."""
    import torch

    q = _as_queue(cq)
    if validate:
        validate_batched_einsum_transform(einsum, q, transform, schedule)
    arg_dict = generate_input_arrays(q, einsum, long_dim_length)
    mode = _placement_mode(transform)
    report: Mapping[str, Any] = {"mode": "separate"}
    if mode == "split":
        from feinsum_amd import placement

        try:
            with torch.cuda.device(q.torch_device):
                out_dict = generate_out_arrays(q, einsum, long_dim_length, split=True)
            infos = {n: placement.split_info(t) for n, t in out_dict.items()}
            report = {"mode": "split", "outputs": {n: ({"pieces_by_class": i["pieces_by_class"], "first_pieces": i["first_pieces"]} if i
                                                      else "torch allocation (below 8 MiB, or no second class of physical memory found)") for n, i in infos.items()},
                      "alloc_ms": round(sum(i.get("alloc_ms", 0.0) for i in infos.values()), 3)}
        except (RuntimeError, HipLibraryError) as exc:     # the allocator could not serve: the reference's protocol, and say so
            logger.warning("split allocator not available (%s); timing torch allocations", str(exc)[:160])
            out_dict = generate_out_arrays(q, einsum, long_dim_length)
            report = {"mode": "separate", "fallback": f"split allocator failed: {str(exc)[:160]}"}
    else:
        out_dict = generate_out_arrays(q, einsum, long_dim_length)
    # `transform={"prepared": True}`: the operator matrices are written once in fragment layout (see
    # _FamilyLaunch.prepare_operators) instead of being rebuilt by every launch
    prepare = _prepared_from_transform(transform, False)
    _, bound, _ = _bind(einsum, q, arg_dict, out_dict, transform, prepare=prepare)
    with torch.cuda.device(q.torch_device):
        for _ in range(N_WARMUP_ROUNDS):
            bound.launch(q.stream_ptr)
        q.finish()
        dev_time = wall_time = 0.0
        rounds = 0
        while rounds < min_rounds or wall_time < min_secs:
            t0 = time()
            dev_time += bound.time_batch(LAUNCHES_PER_BATCH, q.stream_ptr)   # fences like evt.wait()
            wall_time += time() - t0
            rounds += LAUNCHES_PER_BATCH
    return TimingResult(dev_time / rounds, wall_time / rounds, rounds, MappingProxyType(dict(report)))


def timeit(einsum: BatchedEinsum, *, transform: Any = None, cq: Any = None,
           long_dim_length: int = 100000,
           schedule: Optional[ContractionSchedule] = None) -> float:
    """
    Mean seconds per launch of *einsum* on the device of *cq*: validation at
    E = 100, 5 warm-up launches, then batches of 5 launches until at least 10
    launches and 2 s have elapsed (reference: measure.py:197-275).  Device time
    is taken with HIP events around each batch.
    """
    return timeit_details(einsum, transform=transform, cq=cq, long_dim_length=long_dim_length,
                          schedule=schedule).seconds_device


# --------------------------------------------------------------------------
# op counts, footprint, roofline
# --------------------------------------------------------------------------

def _get_giga_ops_from_einsum(expr: BatchedEinsum, long_dim_length: int) -> Mapping[np.dtype, float]:
    """GOps of the optimal schedule by result dtype (reference: measure.py:278-331).

    Complex arithmetic is weighted as in the reference (add = 2, mul = 6 real ops).
    """
    dt = np.result_type(*[np.dtype(d) for d in expr.arg_to_dtype.values()])
    ops = count_ops(expr, long_dim_length=long_dim_length)
    if dt.kind == "c":
        # count_ops returns adds + muls; split them to apply the weights
        from feinsum_amd.contraction_schedule import (_dim_lengths, _step_ops,
                                                      get_opt_einsum_contraction_schedule)
        dims = _dim_lengths(expr, long_dim_length)
        total = 0
        for subs in get_opt_einsum_contraction_schedule(expr).subscripts:
            lhs, rhs = subs.replace(" ", "").split("->")
            sets = [frozenset(s) for s in lhs.split(",")]
            pts = 1
            for idx in frozenset().union(*sets):
                pts *= dims[idx]
            both = _step_ops(sets, frozenset(rhs), dims)
            mults = (len(sets) - 1) * pts
            total += 6 * mults + 2 * (both - mults)
        ops = total * expr.b
        dt = get_real_dtype(dt)
    return MappingProxyType({np.dtype(dt): ops * 1e-9})


def _get_footprint_gbytes(expr: BatchedEinsum, long_dim_length: int) -> float:
    """Every distinct input once + every output once (reference: measure.py:334-354)."""
    nbytes = 0
    for name, shape in expr.arg_to_shape.items():
        nbytes += int(np.prod(_concrete_shape(shape, long_dim_length), dtype=np.int64)) \
            * np.dtype(expr.arg_to_dtype[name]).itemsize
    out_entries = int(np.prod(_concrete_shape(expr.shape, long_dim_length), dtype=np.int64))
    for k in range(expr.b):
        nbytes += out_entries * result_dtype(expr, k).itemsize
    return nbytes * 1e-9


def measure_giga_op_rate(expr: BatchedEinsum, *, transform: Any = None, cq: Any = None,
                         long_dim_length: int = 100000,
                         schedule: Optional[ContractionSchedule] = None) -> Mapping[np.dtype, float]:
    """GOps/s by result dtype (reference: measure.py:357-385)."""
    runtime = timeit(expr, transform=transform, cq=cq, long_dim_length=long_dim_length,
                     schedule=schedule)
    return MappingProxyType({dt: gops / runtime
                             for dt, gops in _get_giga_ops_from_einsum(expr, long_dim_length).items()})


def get_roofline_flop_rate(expr: BatchedEinsum, dev_name: str,
                           long_dim_length: int = 100_000) -> Mapping[np.dtype, float]:
    """
    ``t_roof = max(GOps / peak_GOps[dtype], GB / peak_BW)``; returns GOps / t_roof
    per dtype (reference: measure.py:388-418).  Unknown device ->
    :class:`NoDevicePeaksInfoError`.
    """
    from feinsum_amd.device_info import DEV_TO_PEAK_BW, DEV_TO_PEAK_GFLOPS, normalize_device_name

    dev_name = normalize_device_name(dev_name)
    gops = _get_giga_ops_from_einsum(expr, long_dim_length)
    ngbs = _get_footprint_gbytes(expr, long_dim_length)
    try:
        t_flops = max(g / DEV_TO_PEAK_GFLOPS[dev_name][dt.name] for dt, g in gops.items())
        t_bw = ngbs / DEV_TO_PEAK_BW[dev_name]
    except KeyError as exc:
        raise NoDevicePeaksInfoError(dev_name) from exc
    t_roof = max(t_flops, t_bw)
    return MappingProxyType({dt: g / t_roof for dt, g in gops.items()})


def _strify_measured_vs_roofline(measured: Mapping[np.dtype, Any],
                                 roofline: Mapping[np.dtype, Any]) -> str:
    from tabulate import tabulate

    assert set(measured) == set(roofline)
    fmt = lambda v: f"{v:.1f}" if isinstance(v, float) else str(v)  # noqa: E731
    table = [["Dtype", "Measured GOps/s", "Roofline GOps/s"]]
    for dt in sorted(measured, key=lambda d: d.itemsize):
        table.append([dt.name, fmt(measured[dt]), fmt(roofline[dt])])
    return tabulate(table, tablefmt="fancy_grid")


def stringify_comparison_vs_roofline(expr: BatchedEinsum, *,
                                     schedule: Optional[ContractionSchedule] = None,
                                     transform: Any = None, cq: Any = None,
                                     long_dim_length: int = 100000,
                                     ignore_unknown_device: bool = True) -> str:
    """The reference's fancy_grid table "Dtype / Measured GOps/s / Roofline GOps/s"
    (reference: measure.py:484-525); roofline at the SAME long_dim_length."""
    q = _as_queue(cq)
    measured = measure_giga_op_rate(expr, transform=transform, schedule=schedule, cq=q,
                                    long_dim_length=long_dim_length)
    try:
        roofline: Mapping[np.dtype, Any] = get_roofline_flop_rate(expr, q.device.name, long_dim_length)
    except NoDevicePeaksInfoError:
        if not ignore_unknown_device:
            raise
        roofline = dict.fromkeys(measured, "N/A")
    return _strify_measured_vs_roofline(measured, roofline)
