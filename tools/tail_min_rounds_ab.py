"""Launch times around the switch between the static and the dynamic walk (four and a half rounds: feinsum_hip.hip, tail_static_tiles).
   python tools/tail_min_rounds_ab.py          (profiles/r04/dynamic_walk_from_four_and_a_half_rounds.txt was taken with an
   experiment knob, FEINSUM_TAIL_MIN_ROUNDS, that the rule has since replaced: the label it prints is that variable)"""
import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch
import dg
import feinsum_amd as f
from feinsum_amd import measure
for name, expr in (("grad", dg.grad()), ("div", dg.div()), ("face_mass", dg.face_mass(4))):
    for E in (100_000, 120_000, 131_072, 140_000, 150_000, 163_000):
        t = min(measure.timeit_details(expr, cq=0, long_dim_length=E, min_secs=0.2).seconds_device for _ in range(3))
        print(f"min_rounds={os.environ.get('FEINSUM_TAIL_MIN_ROUNDS','5')} {name} E={E}: {t*1e6:.2f} us", flush=True)
