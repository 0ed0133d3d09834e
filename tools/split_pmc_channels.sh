#!/bin/bash
# Per-L2-channel counters (unsummed: 16 channels x 8 XCDs) of the slow and the fast placement of tools/split_pmc.py.
#   bash tools/split_pmc_channels.sh "<counters>" <tag>       -> gpurun_out/split_pmc_channels/<tag>.json
set -e
repo=$PWD; out=/tmp/split_pmc_ch; keep=$PWD/gpurun_out/split_pmc_channels
rm -rf "$out" && mkdir -p "$out" "$keep"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --pmc $1 --kernel-include-regex facemass --kernel-trace --output-format json -d "$out/p" -o pmc -- \
    python3 "$repo/tools/split_pmc.py" 70 > "$keep/$2.out" 2> "$out/err.txt" || { tail -5 "$out/err.txt"; exit 1; }
f=$(find "$out/p" -name "*.json" | head -1)
ls -la "$f"
python3 - "$f" "$keep/$2.summary.txt" <<'PY'
import json, sys, collections
d = json.load(open(sys.argv[1]))["rocprofiler-sdk-tool"][0]
# counter id -> (name, dims) from the metadata
info = {}
for c in d["counters"]:
    info[c["id"]["handle"]] = c
dims_of = {}
recs = d["callback_records"].get("counter_collection") or d["buffer_records"].get("counter_collection")
print("dispatches:", len(recs), file=sys.stderr)
out = open(sys.argv[2], "w")
# every record: {"dispatch_data": ..., "records": [{"counter_id": {"handle"}, "value"}...]} ; print the raw structure of the first
print(json.dumps(recs[0], indent=None)[:1500], file=out)
print(json.dumps(d["counters"][0])[:800], file=out)
PY
cp "$f" "$keep/$2.json" 2>/dev/null || true
ls -la "$keep"
