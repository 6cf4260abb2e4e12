#!/bin/bash
# HBM-side traffic of the five kernels of one 512^3 apply from PMC counters, as MI355X_MICROARCH.md (HBM section) prescribes:
# separate --pmc passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass), FETCH_SIZE doubled (gfx950 tallies a 128-B
# request at 64 B), WRITE_SIZE as is; both counters are in KB.  Also a --kernel-trace --stats pass of bench.py.
# usage (GPU box, repo root): bash tools/pmc_traffic.sh <tag>      -> gpurun_out/<tag>_*.{json,csv}
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python3 $R/tools/one_apply.py 512 3 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python3 $R/tools/one_apply.py 512 3 > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity-gate > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
python3 - <<EOF
import csv, glob, json, collections
def per_launch(dirname, counter):
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % dirname, recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter: continue
            k = row["Kernel_Name"]
            acc[k] += float(row["Counter_Value"]); 
            if row["Dispatch_Id"] not in seen: seen.add(row["Dispatch_Id"]); cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}
stage = lambda k: next((s for s in ("xfwd", "yfwd", "zfused", "yinv", "xinv") if "k_" + s in k), None)
fetch, write = per_launch("fetch", "FETCH_SIZE"), per_launch("write", "WRITE_SIZE")
N, C = 512 ** 3, 16.0
alg = {"xfwd": N * (C + 8) + 2 * N * C, "yfwd": 6 * N * C, "zfused": 16 * N * C, "yinv": 6 * N * C, "xinv": 4 * N * C}
out, latest = {}, {}
for k in fetch:
    s = stage(k)
    if not s: continue
    rd, wr = 2.0 * fetch[k] * 1024.0, write.get(k, 0.0) * 1024.0
    out[s] = {"kernel": k.split("(")[0][:160], "read_bytes_per_launch (FETCH_SIZE KB x 2 x 1024)": rd, "write_bytes_per_launch (WRITE_SIZE KB x 1024)": wr,
              "traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg[s], "traffic_over_algorithmic": (rd + wr) / alg[s]}
    latest[s] = rd + wr
json.dump({"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 tools/one_apply.py 512 3; gfx950 corrections per MI355X_MICROARCH.md",
           "kernels": out}, open("$OUT/pmc_traffic.json", "w"), indent=1)
import sys; sys.path.insert(0, "$R/tools"); from csrc_hash import csrc_hash
latest["_csrc_sha256"] = csrc_hash("$R")           # bench.py reports the traffic only for the build it was measured on
json.dump(latest, open("$OUT/traffic_latest.json", "w"), indent=1)
print(json.dumps(out, indent=1))
EOF
for f in $(find $OUT/stats -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
