#!/usr/bin/env python
"""Prologue phases per half of the grid from a stamps CSV of fe_check_exp (FE_DUMP_STAMPS):  python tools/phase_report.py <stamps.csv>"""
import csv, statistics as st, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for half, name in ((0, "older"), (1, "younger")):
    ws = [r for r in rows if (int(r["wave"]) // 4) // 256 == half]
    def m(k):
        v = [float(r[k]) for r in ws if float(r[k]) >= 0]
        return f"{st.mean(v):5.2f} (max {max(v):5.2f})" if v else "  -  "
    print(f"{name:8s} entry {m('entry_us')} | operator staged {m('op_landed_us')} | barrier 1 {m('barrier1_us')} | fragments built {m('frags_built_us')} | barrier 2 {m('barrier2_us')} | loop start {m('loop_start_us')} | loop end {m('loop_end_us')}")
