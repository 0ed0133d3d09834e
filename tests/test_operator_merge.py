"""operator._merge: stages share a persistent launch only when no data dependency forbids it
(the bodies of a fused launch run in turn with no grid-wide barrier between them)."""

from types import SimpleNamespace

import pytest

from feinsum_amd import operator
from feinsum_amd.family import FAMILY_DIV, FAMILY_FACEMASS, FAMILY_GRAD, KernelPlan
from feinsum_amd.measure import _FamilyLaunch

MB = 1 << 20
J, D = (1 * MB, 72), (2 * MB, 29400)


def stage(family, reads, writes, b=1):
    """A bound launch as `_bind` leaves it, without touching a device."""
    s = object.__new__(_FamilyLaunch)
    s.plan = KernelPlan(family, 0, {}, "e", {"Np": 35})
    s.variant = 0
    g = SimpleNamespace(J=J[0], D=D[0], E=1000, Np=35, b=b, ndim=3, u=reads[-1][0], out=writes[0][0],
                        nf=4, Nfp=15, layout_flags=0, v=None, outs=None, prepared=None)
    s.groups = [g]
    s.group_family = family
    s.reads, s.writes = tuple(reads), tuple(writes)
    return s


def names(launches):
    return [getattr(b, "entry_point", None) or b.plan.name for b in launches]


def test_independent_div_and_grad_share_a_launch():
    grad = stage(FAMILY_GRAD, [J, D, (10 * MB, 1000)], [(20 * MB, 3000)])
    div = stage(FAMILY_DIV, [J, D, (30 * MB, 3000)], [(40 * MB, 1000)])
    assert names(operator._merge([div, grad])) == ["fe_graddiv3d_f64"]
    lift = stage(FAMILY_FACEMASS, [(50 * MB, 100), (51 * MB, 100), (52 * MB, 4000)], [(60 * MB, 4000)], b=4)
    merged = operator._merge([div, grad, lift])
    assert names(merged) == ["fe_waveop3d_f64"]
    assert set(merged[0].writes) == {(20 * MB, 3000), (40 * MB, 1000), (60 * MB, 4000)}


def test_laplacian_is_not_fused():
    """div of the gradient just computed: the div body would read grad_out before it is written."""
    grad = stage(FAMILY_GRAD, [J, D, (10 * MB, 1000)], [(20 * MB, 3000)])
    div = stage(FAMILY_DIV, [J, D, (20 * MB, 3000)], [(40 * MB, 1000)])
    assert names(operator._merge([grad, div])) == ["grad", "div"]
    # partial overlap counts too
    div2 = stage(FAMILY_DIV, [J, D, (20 * MB + 2992, 3000)], [(40 * MB, 1000)])
    assert names(operator._merge([grad, div2])) == ["grad", "div"]
    # and so does an output written twice, or an input overwritten by the partner
    div3 = stage(FAMILY_DIV, [J, D, (30 * MB, 3000)], [(20 * MB + 8, 1000)])
    assert names(operator._merge([grad, div3])) == ["grad", "div"]
    div4 = stage(FAMILY_DIV, [J, D, (30 * MB, 3000)], [(10 * MB, 1000)])
    assert names(operator._merge([grad, div4])) == ["grad", "div"]


def test_merge_does_not_hop_over_a_producer_or_consumer():
    """grad, X, div with X writing div's input (or reading grad's output while div overwrites it):
    fusing would move div in front of X."""
    grad = stage(FAMILY_GRAD, [J, D, (10 * MB, 1000)], [(20 * MB, 3000)])
    div = stage(FAMILY_DIV, [J, D, (30 * MB, 3000)], [(40 * MB, 1000)])
    producer = stage(FAMILY_FACEMASS, [(50 * MB, 100), (51 * MB, 100), (52 * MB, 4000)], [(30 * MB, 3000)], b=5)
    assert names(operator._merge([grad, producer, div])) == ["grad", "facemass", "div"]
    reader = stage(FAMILY_FACEMASS, [(50 * MB, 100), (51 * MB, 100), (40 * MB, 1000)], [(70 * MB, 4000)], b=5)
    assert names(operator._merge([grad, reader, div])) == ["grad", "facemass", "div"]
    bystander = stage(FAMILY_FACEMASS, [(50 * MB, 100), (51 * MB, 100), (52 * MB, 4000)], [(70 * MB, 4000)], b=5)
    assert names(operator._merge([grad, bystander, div])) == ["fe_graddiv3d_f64", "facemass"]


def test_a_dependent_lift_stays_its_own_launch():
    grad = stage(FAMILY_GRAD, [J, D, (10 * MB, 1000)], [(20 * MB, 3000)])
    div = stage(FAMILY_DIV, [J, D, (30 * MB, 3000)], [(40 * MB, 1000)])
    lift = stage(FAMILY_FACEMASS, [(50 * MB, 100), (51 * MB, 100), (40 * MB, 1000)], [(60 * MB, 4000)], b=4)
    assert names(operator._merge([div, grad, lift])) == ["fe_graddiv3d_f64", "facemass"]


@pytest.mark.parametrize("a,b,hit", [((0, 8), (8, 8), False), ((0, 9), (8, 8), True), ((8, 8), (0, 9), True),
                                     ((0, 0), (0, 8), False), ((4, 2), (0, 8), True)])
def test_overlap(a, b, hit):
    assert operator._overlap([a], [b]) is hit
