"""Launch calls against HIP-graph replay for a multi-launch operator at small element counts.

    python tools/bench_graph.py
"""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import dg  # noqa: E402
import feinsum_amd as f  # noqa: E402

for E in (2_000, 20_000, 100_000, 1_000_000):
    exprs = [dg.div(), dg.grad(), dg.face_mass(5), dg.mass_apply(4)]
    q = f.DeviceQueue(0)
    arrays = [dict(f.generate_input_arrays(q, e, E, np_seed=k)) for k, e in enumerate(exprs)]
    op = f.bind_operator(list(zip(exprs, arrays)), q, fuse=False).capture()
    for _ in range(3):
        op.launch()
        op.replay()
    q.finish()
    n = 200
    t_calls = min(op.time_batch(n) for _ in range(3)) / n
    t_graph = min(op.time_batch(n, graph=True) for _ in range(3)) / n
    print(f"E={E:8d}: {len(op.entry_points)} launches  calls {t_calls * 1e6:8.1f} us   graph replay {t_graph * 1e6:8.1f} us")
