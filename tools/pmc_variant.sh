#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one 512^3 apply under a tuning environment (e.g. LSFC_Z_PERSIST=5), per kernel.
# usage (GPU box, repo root): LSFC_Z_PERSIST=5 bash tools/pmc_variant.sh <tag> [n]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-variant}
N=${2:-512}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python3 $R/tools/one_apply.py $N 2 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python3 $R/tools/one_apply.py $N 2 > $OUT/write.log 2>&1
python3 - <<EOF
import csv, glob, collections
def per_launch(dirname, counter):
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % dirname, recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter: continue
            k = row["Kernel_Name"].split("<")[0]
            acc[k] += float(row["Counter_Value"])
            if row["Dispatch_Id"] not in seen: seen.add(row["Dispatch_Id"]); cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}
f, w = per_launch("fetch", "FETCH_SIZE"), per_launch("write", "WRITE_SIZE")
with open("$OUT/summary.txt", "w") as out:
    for k in f:
        if "k_" not in k: continue
        out.write("%s read_GB=%.3f write_GB=%.3f\n" % (k, 2 * f[k] * 1024 / 1e9, w.get(k, 0) * 1024 / 1e9))
print(open("$OUT/summary.txt").read())
EOF
