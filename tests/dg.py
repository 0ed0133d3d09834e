"""The DG-wave einsums of the hot path, spelled with the current feinsum API the
way the reference's own tests spell them (test/test_codegen.py:34-120,
test/test_measure.py:55-81, test/test_loopy_utils.py:34-48)."""

import feinsum_amd as f

NP, NF, NFP = 35, 4, 15


def grad(Np=NP):
    return f.einsum("xre,rij,ej->xei", f.array("J", (3, 3, "E")), f.array("R", (3, Np, Np)),
                    f.array("u", ("E", Np)))


def div(Np=NP):
    return f.einsum("xre,rij,xej->ei", f.array("J", (3, 3, "E")), f.array("R", (3, Np, Np)),
                    f.array("u", (3, "E", Np)))


def face_mass(b=4, Np=NP, nf=NF, Nfp=NFP):
    return f.batched_einsum(
        "ef, fij, fej -> ei",
        [[f.array("J", ("E", nf)), f.array("R", (nf, Np, Nfp)), f.array(f"v{i}", (nf, "E", Nfp))]
         for i in range(b)])


def face_mass_ifj_fe(b=4, Np=NP, nf=NF, Nfp=NFP):
    return f.batched_einsum(
        "ifj,fe,fej->ei",
        [[f.array("L", (Np, nf, Nfp)), f.array("J", (nf, "E")), f.array(f"v{i}", (nf, "E", Nfp))]
         for i in range(b)])


def batched_div_components(Np=NP):
    # test/test_codegen.py:34-66
    return f.batched_einsum(
        "se, sij, ej -> ei",
        [[f.array("J" + c, (3, "E")), f.array("R", (3, Np, Np)), f.array("u" + c, ("E", Np))]
         for c in "xyz"])


GOLDEN_CASES = {
    "grad_p4": grad,
    "div_p4": div,
    "facemass_p4_ef_fij": face_mass,
    "facemass_p4_ifj_fe": face_mass_ifj_fe,
    "batched_div_p4": batched_div_components,
    "grad_p2": lambda: grad(10),
}


def grad_t(Np=NP):
    # transposed operator sibling, D stored [r][j][i]
    return f.einsum("xre,rji,ej->xei", f.array("J", (3, 3, "E")), f.array("R", (3, Np, Np)),
                    f.array("u", ("E", Np)))


def div_t(Np=NP):
    # tuning/impls/xre_rji_xej_to_ei_v1.py
    return f.einsum("xre,rji,xej->ei", f.array("J", (3, 3, "E")), f.array("R", (3, Np, Np)),
                    f.array("u", (3, "E", Np)))


def face_mass_jfi_fe(b=4, Np=NP, nf=NF, Nfp=NFP):
    # tuning/impls/jfi_fe_fej_to_ei.py:46-56
    return f.batched_einsum(
        "jfi,fe,fej->ei",
        [[f.array("L", (Nfp, nf, Np)), f.array("J", (nf, "E")), f.array(f"v{i}", (nf, "E", Nfp))]
         for i in range(b)])


def face_mass_fji(b=4, Np=NP, nf=NF, Nfp=NFP):
    return f.batched_einsum(
        "ef,fji,fej->ei",
        [[f.array("J", ("E", nf)), f.array("R", (nf, Nfp, Np)), f.array(f"v{i}", (nf, "E", Nfp))]
         for i in range(b)])


def batched_grad(b=3, Np=NP, op="rij"):
    # b fields sharing J and the operator: tuning/impls/batched_xre_rij_ej_to_xei.py
    return f.batched_einsum(
        f"xre,{op},ej->xei",
        [[f.array("J", (3, 3, "E")), f.array("R", (3, Np, Np)), f.array(f"u{i}", ("E", Np))]
         for i in range(b)])


def batched_div(b=3, Np=NP, op="rij"):
    # tuning/impls/batched_xre_rij_xej_to_ei.py (_v2, _v3)
    return f.batched_einsum(
        f"xre,{op},xej->ei",
        [[f.array("J", (3, 3, "E")), f.array("R", (3, Np, Np)), f.array(f"u{i}", (3, "E", Np))]
         for i in range(b)])


CROSS_FIELDS = {"ux": ("Jy", "Jz"), "uy": ("Jx", "Jz"), "uz": ("Jx", "Jy"),
                "vx": ("Jy", "Jz"), "vy": ("Jx", "Jz"), "vz": ("Jx", "Jy")}


def cross_product_batch(Np=NP, op="rji", fields=CROSS_FIELDS):
    # the curl-type batch of tuning/impls/re_rji_ej_to_ei_3d_cross_product_v0.py:220-231:
    # every field component is differentiated along the two other directions
    return f.batched_einsum(
        f"re,{op},ej->ei",
        [[f.array(J, (3, "E")), f.array("D", (3, Np, Np)), f.array(u, ("E", Np))]
         for u, js in fields.items() for J in js])


def mass_apply(b=4, Np=NP, op="ij"):
    # per-element factor times a dense operator: tuning/impls/e_ij_ej_to_ei_no_prftch.py:30-38
    return f.batched_einsum(
        f"e,{op},ej->ei",
        [[f.array("J", ("E",)), f.array("D", (Np, Np)), f.array(f"u{i}", ("E", Np))] for i in range(b)])


def operator_apply(Np=NP, op="ij"):
    # tuning/impls/ij_ej_to_ei_no_prftch.py
    return f.einsum(f"{op},ej->ei", f.array("D", (Np, Np)), f.array("u", ("E", Np)))
