"""Family recogniser: DG einsums are matched up to index renaming and operand
order (the job canonicalisation + the transform archive do in the reference)."""

import itertools
import random
import string

import feinsum_amd as f
from feinsum_amd.family import (FAMILY_DIV, FAMILY_DIVCOMP, FAMILY_FACEMASS, FAMILY_GRAD, FM_J_FE, FM_R_IFJ, FM_R_T,
                                OP_TRANSPOSED)

import dg


def test_basic_matches():
    p = f.match_family(dg.grad())
    assert p.family == FAMILY_GRAD and p.params == {"Np": 35, "ndim": 3} and p.long_index == "e"
    assert p.roles == {"J": 0, "D": 1, "u": 2}
    p = f.match_family(dg.div())
    assert p.family == FAMILY_DIV
    p = f.match_family(dg.face_mass())
    assert p.family == FAMILY_FACEMASS and p.layout_flags == 0
    assert p.params == {"Np": 35, "nf": 4, "Nfp": 15}
    p = f.match_family(dg.face_mass_ifj_fe())
    assert p.family == FAMILY_FACEMASS and p.layout_flags == (FM_J_FE | FM_R_IFJ)
    assert p.roles == {"J": 1, "R": 0, "v": 2}
    assert f.match_family(dg.grad(10)).params == {"Np": 10, "ndim": 3}


def test_non_family():
    A = f.array("A", (10, 4), "float32")
    assert f.match_family(f.einsum("ij,j->i", A, f.array("x", 4, "float32"))) is None
    # right structure, wrong output layout (e and i swapped): memory layout differs
    g = f.einsum("xre,rij,ej->xie", f.array("J", (3, 3, "E")), f.array("R", (3, 35, 35)),
                 f.array("u", ("E", 35)))
    assert f.match_family(g) is None
    # all-float32 DG einsum: a family einsum too (round 3: fe_launch_f32), marked so that the float launch is taken
    g32 = f.einsum("xre,rij,ej->xei", f.array("J", (3, 3, "E"), "float32"),
                   f.array("R", (3, 35, 35), "float32"), f.array("u", ("E", 35), "float32"))
    plan = f.match_family(g32)
    assert plan is not None and plan.params == {"Np": 35, "ndim": 3, "f32": 1}
    # mixed element types are not
    gmix = f.einsum("xre,rij,ej->xei", f.array("J", (3, 3, "E"), "float32"),
                    f.array("R", (3, 35, 35), "float64"), f.array("u", ("E", 35), "float32"))
    assert f.match_family(gmix) is None
    # 2D grad (ndim = 2) is the grad family with ndim = 2 (tiled kernel); other ndim are not
    g2 = f.einsum("xre,rij,ej->xei", f.array("J", (2, 2, "E")), f.array("R", (2, 10, 10)),
                  f.array("u", ("E", 10)))
    assert f.match_family(g2).params == {"Np": 10, "ndim": 2}
    g4 = f.einsum("xre,rij,ej->xei", f.array("J", (4, 4, "E")), f.array("R", (4, 10, 10)),
                  f.array("u", ("E", 10)))
    assert f.match_family(g4) is None
    g23 = f.einsum("xre,rij,ej->xei", f.array("J", (2, 3, "E")), f.array("R", (3, 10, 10)),
                   f.array("u", ("E", 10)))
    assert f.match_family(g23) is None


def test_renaming_and_operand_order_fuzz():
    rng = random.Random(0)
    shapes = {"J": (3, 3, "E"), "R": (3, 35, 35), "u": ("E", 35)}
    base = {"J": "xre", "R": "rij", "u": "ej"}
    for _ in range(200):
        letters = rng.sample(string.ascii_lowercase, 5)
        ren = dict(zip("xreij", letters))
        order = rng.sample(["J", "R", "u"], 3)
        subs = ",".join("".join(ren[c] for c in base[o]) for o in order) + "->" + \
            "".join(ren[c] for c in "xei")
        names = {o: "".join(rng.sample(string.ascii_uppercase, 3)) + "_" + o for o in order}
        expr = f.einsum(subs, *[f.array(names[o], shapes[o]) for o in order])
        p = f.match_family(expr)
        assert p is not None and p.family == FAMILY_GRAD, subs
        assert p.long_index == ren["e"]
        assert [order[p.roles[r]] for r in ("J", "D", "u")] == ["J", "R", "u"]


def test_all_operand_orders_div():
    for order in itertools.permutations(range(3)):
        ops = [("xre", f.array("J", (3, 3, "E"))), ("rij", f.array("R", (3, 35, 35))),
               ("xej", f.array("u", (3, "E", 35)))]
        subs = ",".join(ops[k][0] for k in order) + "->ei"
        p = f.match_family(f.einsum(subs, *[ops[k][1] for k in order]))
        assert p is not None and p.family == FAMILY_DIV
        assert order[p.roles["J"]] == 0 and order[p.roles["D"]] == 1 and order[p.roles["u"]] == 2


def test_transposed_operator_siblings():
    p = f.match_family(dg.grad_t())
    assert p.family == FAMILY_GRAD and p.layout_flags == OP_TRANSPOSED
    p = f.match_family(dg.div_t())
    assert p.family == FAMILY_DIV and p.layout_flags == OP_TRANSPOSED
    assert f.match_family(dg.grad()).layout_flags == 0 and f.match_family(dg.div()).layout_flags == 0
    p = f.match_family(dg.face_mass_jfi_fe())
    assert p.family == FAMILY_FACEMASS and p.layout_flags == (FM_J_FE | FM_R_IFJ | FM_R_T)
    assert p.roles == {"J": 1, "R": 0, "v": 2} and p.params == {"Np": 35, "nf": 4, "Nfp": 15}
    p = f.match_family(dg.face_mass_fji())
    assert p.family == FAMILY_FACEMASS and p.layout_flags == FM_R_T


def test_div_component_family():
    # test/test_codegen.py:34-66 'se, sij, ej -> ei' x 3, and the 'es' layout of examples/dg_wave_div.py
    p = f.match_family(dg.batched_div_components())
    assert p.family == FAMILY_DIVCOMP and p.layout_flags == 0 and p.params == {"Np": 35, "ndim": 3}
    es = f.batched_einsum("es,sij,ej->ei", [[f.array("J" + c, ("E", 3)), f.array("R", (3, 35, 35)),
                                             f.array("u" + c, ("E", 35))] for c in "xyz"])
    assert f.match_family(es).layout_flags == 2
    # triangles (s = 2) are the same family with ndim = 2; other component counts are not DG einsums
    e2 = f.einsum("se,sij,ej->ei", f.array("J", (2, "E")), f.array("R", (2, 10, 10)), f.array("u", ("E", 10)))
    assert f.match_family(e2).params == {"Np": 10, "ndim": 2}
    e4 = f.einsum("se,sij,ej->ei", f.array("J", (4, "E")), f.array("R", (4, 10, 10)), f.array("u", ("E", 10)))
    assert f.match_family(e4) is None
