#!/usr/bin/env python
"""Per-(XCD, L2 channel) means of unsummed TCC counters from a rocprofv3 JSON of tools/split_pmc.py
(last 120 dispatches = 3 configurations x 40).   python tools/pmc_channels_table.py <json> [counter ...]"""
import json
import sys
import collections

d = json.load(open(sys.argv[1]))["rocprofiler-sdk-tool"][0]
names = {c["id"]["handle"]: c["name"] for c in d["counters"]}
recs = d["callback_records"].get("counter_collection") or d["buffer_records"].get("counter_collection")
recs = sorted(recs, key=lambda r: r["dispatch_data"]["dispatch_info"]["dispatch_id"])[-120:]
cfgs = ["4 below", "2 + 2", "4 above"]
want = sys.argv[2:] or sorted(set(names.values()))
for ci, cfg in enumerate(cfgs):
    chunk = recs[ci * 40 + 10:(ci + 1) * 40]
    acc = collections.defaultdict(lambda: [0.0] * 128)
    dur = 0.0
    for r in chunk:
        per = collections.defaultdict(list)
        for x in r["records"]:
            per[names[x["counter_id"]["handle"]]].append(x["value"])
        for n, v in per.items():
            assert len(v) == 128, (n, len(v))
            for i, val in enumerate(v):
                acc[n][i] += val / len(chunk)
        dur += (r["dispatch_data"]["end_timestamp"] - r["dispatch_data"]["start_timestamp"]) / len(chunk) * 1e-3
    print(f"=== {cfg}: {dur:.1f} us")
    for n in want:
        v = acc[n]
        print(f"{n}: total {sum(v):.4g}; per channel (mean over the 8 XCDs): " +
              " ".join(f"{sum(v[x * 16 + ch] for x in range(8)) / 8:.3g}" for ch in range(16)))
        print("   per XCD (sum over its 16 channels): " + " ".join(f"{sum(v[x * 16:(x + 1) * 16]):.3g}" for x in range(8)))
        mx = max(v); mn = min(v)
        print(f"   min {mn:.4g}  max {mx:.4g}  max/mean {mx / (sum(v) / 128):.2f}")
