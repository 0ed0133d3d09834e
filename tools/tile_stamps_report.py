#!/usr/bin/env python
"""
Report on the per-wave, per-tile time stamps of ONE launch (experiments build; fe_check_exp ab ... with FE_DUMP_STAMPS):

    python tools/tile_stamps_report.py <stamps.csv.tiles.csv> <grad|div> [bytes_per_tile_stored]

Stamps per tile (microseconds after the first wave's entry): grad {loads landed, matrix work issued, stores issued};
div {loads landed, B fragments built + next loads issued, matrix work issued, stores issued}.  Printed: when the phases of
the k-th tile begin and how long they last (older half = blocks 0..255 / younger half of the grid), how busy the matrix pipes
are over time (a SIMD holds one wave of either half: HW_ID), and the chip-wide store-issue rate over time.
"""
import csv
import statistics as st
import sys

path, fam = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
NPH = 3 if fam == "grad" else 4
MFMA = (0, 1) if fam == "grad" else (1, 2)        # the matrix phase lies between these two stamps
STORE = (1, 2) if fam == "grad" else (2, 3)       # stage 2 / transposition / store issue
store_bytes = float(sys.argv[3]) if len(sys.argv) > 3 else (3 * 4480.0 if fam == "grad" else 4480.0)

waves = []
for r in rows:
    w = int(r["wave"])
    tiles = int(r["tiles"])
    t = [[float(r[f"t{k}_{p}"]) for p in range(4)] for k in range(4)]
    waves.append({"w": w, "half": (w // 4) // 256, "xcc": int(r["xcc"]), "hw": int(r["hw_id"]), "entry": float(r["entry_us"]),
                  "end": float(r["loop_end_us"]), "tiles": tiles, "t": t})


def col(vals):
    vals = sorted(vals)
    return f"mean {st.mean(vals):6.2f}  p10 {vals[len(vals) // 10]:6.2f}  p50 {vals[len(vals) // 2]:6.2f}  p90 {vals[len(vals) * 9 // 10]:6.2f}  max {vals[-1]:6.2f}"


print(f"# {path}: {len(waves)} waves, tiles per wave {st.mean(w['tiles'] for w in waves):.2f}")
names = ["loads landed", "matrix work issued", "stores issued"] if fam == "grad" else \
        ["loads landed", "B built + next loads issued", "matrix work issued", "stores issued"]
for half in (0, 1):
    print(f"## {'older' if half == 0 else 'younger'} half of the grid")
    for k in range(4):
        ws = [w for w in waves if w["half"] == half and w["tiles"] > k and w["t"][k][0] > 0]
        if not ws:
            continue
        print(f"  tile {k} ({len(ws)} waves)")
        for p in range(NPH):
            print(f"    {names[p]:30s} at  {col([w['t'][k][p] for w in ws])}")
        prev_end = [(w["t"][k][0] - (w["t"][k - 1][NPH - 1] if k else w["entry"])) for w in ws]
        print(f"    {'wait before (from prev. stores)':30s} us  {col(prev_end)}")
        for p in range(1, NPH):
            print(f"    {'-> ' + names[p]:30s} us  {col([w['t'][k][p] - w['t'][k][p - 1] for w in ws])}")
    print(f"  loop end                           at  {col([w['end'] for w in waves if w['half'] == half])}")

# matrix-pipe occupancy over time: intervals [t_a, t_b] of all waves, binned at 0.25 us; 1024 SIMDs
T = max(w["end"] for w in waves)
nb = int(T / 0.25) + 2
busy = [0.0] * nb
stores = [0.0] * nb
for w in waves:
    for k in range(min(w["tiles"], 4)):
        a, b = w["t"][k][MFMA[0]], w["t"][k][MFMA[1]]
        if a <= 0 or b <= 0:
            continue
        i = a
        while i < b:
            j = min(b, (int(i / 0.25) + 1) * 0.25)
            busy[int(i / 0.25)] += (j - i)
            i = j
        s = w["t"][k][STORE[1]]
        stores[int(s / 0.25)] += store_bytes
print("## over time (0.5 us bins): waves in their matrix phase (of 1024 pipes; two waves of a SIMD in the phase at once share the pipe) | store bytes issued, TB/s")
for i in range(0, nb - 1, 2):
    m = (busy[i] + busy[i + 1]) / 0.5
    sb = (stores[i] + stores[i + 1]) / 0.5e-6 * 1e-12
    print(f"  {i * 0.25:5.1f} us  {m:7.0f} waves {'#' * int(m / 32):40s} | {sb:5.2f} TB/s {'*' * int(sb * 4)}")
# per-SIMD matrix time: HW_ID bits -- wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...; with the XCC id a SIMD key
simd = {}
for w in waves:
    key = (w["xcc"], (w["hw"] >> 4) & 3, (w["hw"] >> 8) & 0xff)
    tot = sum(max(0.0, w["t"][k][MFMA[1]] - w["t"][k][MFMA[0]]) for k in range(min(w["tiles"], 4)) if w["t"][k][0] > 0)
    simd.setdefault(key, []).append((w, tot))
print(f"## {len(simd)} distinct (xcc, simd, cu/sh/se) keys; waves per key: {st.mean(len(v) for v in simd.values()):.2f}")
span = [max(w["end"] for w, _ in v) - min(w["t"][0][0] for w, _ in v if w["t"][0][0] > 0) for v in simd.values() if any(w["t"][0][0] > 0 for w, _ in v)]
print(f"   per SIMD: first loads landed -> last loop end   {col(span)}")
