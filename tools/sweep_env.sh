#!/bin/bash
# bench.py under a list of values of one environment variable (GPU box, repo root):
#   bash tools/sweep_env.sh <out log> VAR v1 v2 ...      (other LSFC_* variables are inherited)
set -e
OUT=$1; VAR=$2; shift 2
for v in "$@"; do
  export $VAR=$v
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-parity-gate --steps 40 > /tmp/sw.json 2>/dev/null
  python3 - <<EOF >> $OUT
import json; d = json.load(open("/tmp/sw.json")); print("$VAR=$v", round(d["ms_per_step"], 3), [round(s["ms"], 3) for s in d["roofline"]["stages"]])
EOF
done
