"""
Several einsums of one DG operator evaluation as a single enqueue.

The reference drives the whole 3-D wave operator from one description:
``examples/wave_3d_p4_auto.py:16-63`` holds div(v), grad(u) and the lift of four
face fields in one kernel, matches each tagged part against an einsum
(``get_a_matched_einsum``, ``:69-71``) and applies that einsum's transform
(``:119-139``); the parts stay separated by global barriers, i.e. they run one
after the other.  None of them reads what another one writes.

Here the description is a list of *stages* ``(BatchedEinsum, arrays)``.  Every
stage is bound to its kernel family exactly as :func:`feinsum_amd.evaluate`
does; stages that the library can run inside one persistent launch are then
merged:

* div + grad that share the geometry factors J and the operator D
  (``fe_graddiv3d_f64``; BASELINE config 3),
* div + grad + face-mass of 2..4 fields (``fe_waveop3d_f64``; config 5),

and everything else is enqueued stage by stage in the order given.  Results do
not depend on the merge: a merged launch runs the same per-tile arithmetic, and
stages are merged only when no stage of the set -- and no stage the merge would
move them across -- reads or overwrites what another one writes
(``_can_share_a_launch``).
"""

from __future__ import annotations

from types import MappingProxyType
from typing import Any, List, Mapping, Optional, Sequence, Tuple

from feinsum_amd import _hip
from feinsum_amd.einsum import BatchedEinsum
from feinsum_amd.family import FAMILY_DIV, FAMILY_FACEMASS, FAMILY_GRAD
from feinsum_amd.measure import _bind, _FamilyLaunch

StageT = Tuple[BatchedEinsum, Mapping[str, Any]]


class _GradDivLaunch:
    """div + grad in one launch (shared J and D)."""

    entry_point = "fe_graddiv3d_f64"

    def __init__(self, grad: _FamilyLaunch, div: _FamilyLaunch) -> None:
        self._keep = (grad, div)
        g, d = grad.groups[0], div.groups[0]
        self.args = (g.J, g.D, g.prepared or d.prepared, g.u, d.u, g.out, d.out, g.E, g.Np, grad.variant)

    def launch(self, stream_ptr: int) -> None:
        _hip.check(_hip.load_library().fe_graddiv3d_prepared_f64(*self.args, stream_ptr))


class _WaveOpLaunch:
    """div + grad + face-mass in one launch."""

    entry_point = "fe_waveop3d_f64"

    def __init__(self, grad: _FamilyLaunch, div: _FamilyLaunch, lift: _FamilyLaunch) -> None:
        self._keep = (grad, div, lift)
        g, d, m = grad.groups[0], div.groups[0], lift.groups[0]
        self.args = (g.J, g.D, g.prepared or d.prepared, g.u, g.out, d.u, d.out, m.J, m.D, m.prepared, m.v, m.outs,
                     g.E, g.Np, m.nf, m.Nfp, m.b, m.layout_flags, grad.variant)

    def launch(self, stream_ptr: int) -> None:
        _hip.check(_hip.load_library().fe_waveop3d_prepared_f64(*self.args, stream_ptr))


def _single_plain_group(b: Any, family: int) -> bool:
    return (isinstance(b, _FamilyLaunch) and b.plan.family == family and len(b.groups) == 1
            and (family == FAMILY_FACEMASS
                 or (b.groups[0].b == 1 and b.plan.layout_flags == 0 and b.groups[0].ndim == 3)))


def _overlap(a: Sequence[Tuple[int, int]], b: Sequence[Tuple[int, int]]) -> bool:
    """Do two lists of (address, nbytes) ranges share a byte?"""
    return any(pa < pb + nb and pb < pa + na for pa, na in a for pb, nb in b if na and nb)


def _conflict(x: Any, y: Any) -> bool:
    """Must launches x and y keep their order (one writes what the other reads or writes)?"""
    return (_overlap(x.writes, y.reads) or _overlap(x.reads, y.writes) or _overlap(x.writes, y.writes))


def _can_share_a_launch(bound: List[Any], members: List[Any]) -> bool:
    """May *members* run as one persistent launch placed where the first of them stands?

    The bodies of a fused launch run in turn with only block barriers between them (a block that
    has finished its div tiles starts its grad tiles while other blocks are still in div), so no
    member may read or overwrite what another member writes -- a Laplacian staged as grad followed
    by div OF that gradient must stay two launches.  Moving the later members up to the first
    one's position also hops them over the stages in between: none of those may conflict with a
    member that passes it.
    """
    for k, x in enumerate(members):
        if any(_conflict(x, y) for y in members[k + 1:]):
            return False
    pos = {id(b): k for k, b in enumerate(bound)}
    first = min(pos[id(m)] for m in members)
    mine = {id(m) for m in members}
    for m in members:
        for hopped in bound[first + 1:pos[id(m)]]:
            if id(hopped) not in mine and _conflict(m, hopped):
                return False
    return True


def _merge(bound: List[Any]) -> List[Any]:
    """Replace the first mergeable (div, grad[, face-mass]) set by its fused launch."""
    grads = [b for b in bound if _single_plain_group(b, FAMILY_GRAD)]
    for g in grads:
        gp = g.groups[0]
        for d in bound:
            if not _single_plain_group(d, FAMILY_DIV):
                continue
            dp = d.groups[0]
            if (dp.J, dp.D, dp.E, dp.Np, d.variant) != (gp.J, gp.D, gp.E, gp.Np, g.variant):
                continue
            if not _can_share_a_launch(bound, [g, d]):
                continue
            lift = next((m for m in bound if _single_plain_group(m, FAMILY_FACEMASS)
                         and 2 <= m.groups[0].b <= 4 and m.variant == g.variant
                         and (m.groups[0].E, m.groups[0].Np) == (gp.E, gp.Np)
                         and _can_share_a_launch(bound, [g, d, m])), None)
            fused = _WaveOpLaunch(g, d, lift) if lift is not None else _GradDivLaunch(g, d)
            members = [g, d] + ([lift] if lift is not None else [])
            fused.reads = tuple(r for m in members for r in m.reads)
            fused.writes = tuple(w for m in members for w in m.writes)
            gone = {id(m) for m in members}
            first = min(k for k, b in enumerate(bound) if id(b) in gone)
            rest = [b for b in bound if id(b) not in gone]
            return rest[:first] + [fused] + _merge(rest[first:])
    return bound


class BoundOperator:
    """The stages of an operator bound to device arrays; ``launch`` enqueues all of them."""

    def __init__(self, queue: Any, launches: List[Any], outputs: List[Mapping[str, Any]]) -> None:
        self.queue, self.launches, self.outputs = queue, launches, outputs
        self._graph = None
        self._capture_id = 0

    @property
    def entry_points(self) -> Tuple[str, ...]:
        """C entry point behind every enqueue, in order (one per launch group)."""
        names: List[str] = []
        for b in self.launches:
            if hasattr(b, "entry_point"):
                names.append(b.entry_point)
            elif isinstance(b, _FamilyLaunch):
                names += ["fe_gradplanes" if b.group_family == 6 else f"fe_{b.plan.name}"] * len(b.groups)
            else:
                names += ["fe_einsum_generic"] * len(b.launches)
        return tuple(names)

    def launch(self, stream_ptr: Optional[int] = None) -> None:
        s = self.queue.stream_ptr if stream_ptr is None else stream_ptr
        for b in self.launches:
            b.launch(s)

    def refresh_operators(self) -> None:
        """Re-prepare the operator matrices after their arrays were changed in place (the prepared copies
        are snapshots taken at bind time; the fused launches hold the same buffers)."""
        for b in getattr(self, "_stages", []):
            if isinstance(b, _FamilyLaunch) and b._prepared:
                b.prepare_operators(self.queue.stream_ptr)

    def capture(self) -> "BoundOperator":
        """Record one evaluation of the operator in a HIP graph (the launchers neither allocate nor
        synchronise, so they can be captured); :meth:`replay` then enqueues it with a single call.
        Pays off for operators of several launches on small element counts, where the host-side
        cost of the launches is comparable to the kernels.  The bound arrays stay the operands:
        update them in place between replays.

        ONE executable per capture: a captured launch's ticket counters belong to the graph node (include/feinsum_hip.h), so
        this object holds the one executable of its capture; capturing again first gives the previous graph's counter groups
        back (:meth:`release_graph`), and so does deleting the operator."""
        import torch

        from feinsum_amd import _hip

        self.release_graph()
        with torch.cuda.device(self.queue.torch_device):
            self.launch()                       # warm-up outside capture: kernel attributes
            self.queue.finish()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=torch.cuda.Stream(self.queue.torch_device)):
                s = int(torch.cuda.current_stream().cuda_stream)
                self._capture_id = _hip.capture_id(s)
                self.launch(s)
        self._graph = graph
        return self

    def release_graph(self) -> int:
        """Destroy the captured graph (after waiting for its replays) and hand its launches' ticket-counter groups back to the
        library (``fe_graph_retired``); returns the number of groups returned.  Called by :meth:`capture` and on deletion."""
        if self._graph is None:
            return 0
        import torch

        from feinsum_amd import _hip

        with torch.cuda.device(self.queue.torch_device):
            torch.cuda.synchronize()
            self._graph.reset()
            self._graph = None
            cid, self._capture_id = self._capture_id, 0
            return _hip.graph_retired(cid) if cid else 0

    def __del__(self) -> None:
        try:
            self.release_graph()
        except Exception:       # noqa: BLE001  (interpreter shutdown: the library or torch may be gone)
            pass

    def replay(self) -> None:
        """Enqueue the captured evaluation on the current stream."""
        if self._graph is None:
            raise RuntimeError("capture() the operator before replay()")
        self._graph.replay()

    def time_batch(self, n: int, stream_ptr: Optional[int] = None, *, graph: bool = False) -> float:
        """Seconds for *n* evaluations of the whole operator (events on the launch stream);
        ``graph=True`` times replays of the captured graph instead of the launch calls."""
        import torch

        if graph:
            stream = torch.cuda.current_stream(self.queue.torch_device)
            step = self.replay
        else:
            s = self.queue.stream_ptr if stream_ptr is None else stream_ptr
            stream = torch.cuda.ExternalStream(s) if s else torch.cuda.current_stream()
            step = lambda: self.launch(s)   # noqa: E731
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(stream)
        for _ in range(n):
            step()
        t1.record(stream)
        t1.synchronize()
        return t0.elapsed_time(t1) * 1e-3


def bind_operator(stages: Sequence[StageT], cq: Any, *,
                  out_dicts: Optional[Sequence[Optional[Mapping[str, Any]]]] = None,
                  transform: Any = None, fuse: bool = True, prepare: bool = False) -> BoundOperator:
    """Bind every stage (shape / dtype / device checks as in ``evaluate``) and merge what can share a launch.

    *prepare* (off by default: the bound arrays stay the operands, whatever is written into them between
    launches): write the operator matrices once in the kernels' fragment layout
    (``fe_prepare_operator``), so that the launches of this bound operator skip rebuilding them -- for
    operators that stay constant across launches, as in a time integrator.  The prepared copies are
    snapshots: after changing an operator array in place call :meth:`BoundOperator.refresh_operators`."""
    if out_dicts is not None and len(out_dicts) != len(stages):
        raise ValueError("out_dicts: need one entry (or None) per stage")
    queue, bound, outputs = None, [], []
    for k, (expr, arrays) in enumerate(stages):
        q, b, outs = _bind(expr, cq, arrays, None if out_dicts is None else out_dicts[k], transform, prepare=prepare)
        queue = queue or q
        bound.append(b)
        outputs.append(MappingProxyType(dict(zip(expr.output_names, outs))))
    op = BoundOperator(queue, _merge(bound) if fuse else bound, outputs)
    op._stages = bound
    return op


def evaluate_operator(stages: Sequence[StageT], cq: Any, *,
                      out_dicts: Optional[Sequence[Optional[Mapping[str, Any]]]] = None,
                      transform: Any = None, fuse: bool = True, wait: bool = False) -> List[Mapping[str, Any]]:
    """Enqueue all stages; returns one ``{"_fe_out": tensor, ...}`` mapping per stage."""
    import torch

    op = bind_operator(stages, cq, out_dicts=out_dicts, transform=transform, fuse=fuse)
    if op.queue is not None:
        with torch.cuda.device(op.queue.torch_device):
            op.launch()
        if wait:
            op.queue.finish()
    return op.outputs
