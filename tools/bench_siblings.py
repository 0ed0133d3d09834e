import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, feinsum_amd as f, dg
for name, expr in (("batched div components x3 (se,sij,ej->ei)", dg.batched_div_components()),
                   ("cross-product batch x12 (re,rji,ej->ei)", dg.cross_product_batch()),
                   ("div transposed (xre,rji,xej->ei)", dg.div_t()), ("face-mass jfi,fe (x4)", dg.face_mass_jfi_fe())):
    r = f.timeit_details(expr, cq=0, long_dim_length=1_000_000, min_secs=0.3)
    gops = f.count_ops(expr, long_dim_length=1_000_000) * 1e-9
    roof = f.get_roofline_flop_rate(expr, "AMD Instinct MI355X", 1_000_000)[np.dtype("float64")]
    print(f"{name}: {r.seconds_device*1e3:.4f} ms  {gops/r.seconds_device:.0f} GFLOP/s  roofline {roof:.0f}  -> {gops/r.seconds_device/roof*100:.1f} %")
