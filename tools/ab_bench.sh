#!/bin/bash
# A/B of tuning environments through bench.py (GPU box, repo root): bash tools/ab_bench.sh <out log> VAR v1 v2 [rounds]
# every run must succeed (set -e): a failing GPU step ends the script
set -e
OUT=$1; VAR=$2; A=$3; B=$4; R=${5:-2}
for ((i = 0; i < R; ++i)); do
  for v in $A $B; do
    export $VAR=$v
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-parity-gate --steps 40 > /tmp/ab.json 2>/dev/null
    python3 - <<EOF >> $OUT
import json; d = json.load(open("/tmp/ab.json")); print("$VAR=$v", round(d["ms_per_step"], 3), [round(s["ms"], 3) for s in d["roofline"]["stages"]])
EOF
  done
done
