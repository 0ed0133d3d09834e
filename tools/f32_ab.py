"""A/B of the float32 kernels' walk (static / tickets) and load hint (non-temporal / plain) on one box.
   python tools/f32_ab.py [grad|div|face_mass ...]"""
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import dg
import feinsum_amd as f
from feinsum_amd import _hip, measure


def f32(expr):
    return f.batched_einsum(expr.get_subscripts(), [[f.array(a.name, a.shape, "float32") for a in row] for row in expr.args])


CASES = {"grad": (dg.grad, 596, 7980), "div": (dg.div, None, None), "face_mass": (lambda: dg.face_mass(4), None, None),
         "grad_p3": (lambda: dg.grad(20), None, None), "grad_p2": (lambda: dg.grad(10), None, None), "grad_p1": (lambda: dg.grad(4), None, None)}
names = sys.argv[1:] or ["grad"]
for name in names:
    expr = f32(CASES[name][0]())
    for E in (400_000, 1_000_000, 2_000_000, 4_000_000):
        row = []
        for rounds, mib in ((-1, 0), (-1, 248), (1 << 20, 0), (1 << 20, 248)):
            _hip.set_tail_rounds(rounds)
            _hip.set_temporal_loads_mib(mib)
            best = min(measure.timeit_details(expr, cq=0, long_dim_length=E, min_secs=0.25).seconds_device for _ in range(3))
            row.append(best * 1e6)
        gf = measure._get_giga_op_count(expr, E) if hasattr(measure, "_get_giga_op_count") else None
        print(f"{name} float32 E={E}: static/nt {row[0]:.2f}  static/plain {row[1]:.2f}  tickets/nt {row[2]:.2f}  tickets/plain {row[3]:.2f} us", flush=True)
