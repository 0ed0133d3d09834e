#!/bin/bash
# Run a list of GPU steps in one gpurun call: each step's output goes to gpurun_out/<tag>/<name>.log,
# an ordinary failure is recorded and the next step runs; a step that times out or is killed ends
# the session (no further GPU work after a hang).
#   bash tools/gpu_session.sh <tag> "<name>|<timeout s>|<command>" ...
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
: > "$out/session.txt"
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; tmo=${rest%%|*}; cmd=${rest#*|}
  echo "[$(date +%H:%M:%S)] $name: $cmd" | tee -a "$out/session.txt"
  t0=$(date +%s)
  timeout -k 10 "$tmo" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "[$(date +%H:%M:%S)] $name: rc=$rc in $(( $(date +%s) - t0 )) s" | tee -a "$out/session.txt"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
    echo "step $name timed out or was killed: stopping the session" | tee -a "$out/session.txt"
    exit 1
  fi
done
exit 0
