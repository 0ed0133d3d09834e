"""grad at E ~ 1e5: launch time against the number of 16-element tiles per SIMD (1024 SIMDs, two waves each).
   python tools/grad_small_quantization.py"""
import sys
from pathlib import Path
root = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(root / "tests")); sys.path.insert(0, str(root))
import torch  # noqa: F401
import dg
import feinsum_amd as f
from feinsum_amd import measure

expr = dg.grad()
for tiles in (6144, 6160, 6250, 6656, 7168, 7184, 8192, 8208):
    E = tiles * 16
    t = min(measure.timeit_details(expr, cq=0, long_dim_length=E, min_secs=0.25).seconds_device for _ in range(3))
    flops = 7980 * E     # 2 * 3 * 35 * 35 + 3 * 35 * 6 - ... (counted as the reference does: see below)
    gb = 1192 * E
    print(f"E={E:7d} tiles={tiles} ({tiles / 1024:.2f} per SIMD, most loaded SIMD {-(-tiles // 1024) if tiles % 1024 == 0 else tiles // 2048 * 2 + (1 if tiles % 2048 <= 1024 else 2)}?): {t * 1e6:6.2f} us  "
          f"{gb / t / 1e12:.3f} TB/s = {gb / t / 8e12:.3f} of the roofline  ({t * 1e6 / (tiles / 1024):.2f} us per tile-per-SIMD)", flush=True)
